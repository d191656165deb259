"""CPU tier: the pure-Python parts of the profiling / ISA tooling."""
import os
import textwrap
from dataclasses import replace

from flash_attention_from_scratch_amd.tools import isa_lint64, isa_stats, kernel_resources, rocprof_bench
from flash_helpers import kernel_configs as kc
from tests.conftest import ROOT

SAMPLE_ASM = textwrap.dedent("""\
    \t.text
    _ZN2fa13fa_fwd_kernelILi15ELi1ELi8ELi64ELb1ELb1ELb0ELb1ELi0EEEvNS_10KernelArgsE: ; @k
    \ts_load_dwordx2 s[0:1], s[4:5], 0x0
    .LBB0_1:                                ; =>This Inner Loop Header: Depth=1
    \ts_waitcnt lgkmcnt(7)
    \tv_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]
    \tds_read_b128 v[24:27], v28
    \tds_read_b64_tr_b16 v[30:31], v29
    \tv_exp_f32_e32 v40, v41
    \tv_fmamk_f32 v42, v43, 0x3e000000, v44
    \ts_barrier
    \ts_cbranch_scc1 .LBB0_1
    \tglobal_store_dwordx2 v[50:51], v[52:53], off
    \ts_endpgm
    """)


def test_isa_stats_histogram_and_trace():
    ks = isa_stats.kernels(SAMPLE_ASM)
    assert len(ks) == 1
    (name, lines), = ks.items()
    loop = isa_stats.hot_loop(lines)
    ops = isa_stats.histogram(loop)
    assert ops["v_mfma_f32_32x32x16_bf16"] == 1 and ops["ds_read_b128"] == 1 and "global_store_dwordx2" not in ops
    cls = isa_stats.class_summary(ops)
    assert cls["mfma"] == 1 and cls["lds"] == 2 and cls["trans"] == 1 and cls["barrier"] == 1
    seq = "".join(isa_stats.classify(*isa_stats.opcode(l)) for l in loop if isa_stats.opcode(l)[0])
    assert seq == "<L7>MdtEv|B|J"


def test_symbol_and_mangled_name_round_trip_to_config():
    cfg = rocprof_bench.symbol_to_config("void fa::fa_fwd_kernel<15, 1, 8, 64, true, true, false, true, true, false, 128, 0>(fa::KernelArgs)")
    assert cfg == kc.FlashForwardKernelConfig(kc.DType.BF16, 128, 256, 64, 8, True, True, True, 0, 0, 0, True, False)
    cfg16 = rocprof_bench.symbol_to_config("void fa::fa_fwd_kernel16<5, 4, 32, true, true, true>(fa::KernelArgs)")
    # (OPT on a double-buffered LDS-DMA variant builds the speculative softmax: a native config, optimized_softmax unset)
    assert (cfg16.dtype, cfg16.B_r, cfg16.B_c, cfg16.n_warps, cfg16.optimized_softmax, cfg16.speculative_softmax) == \
        (kc.DType.FP16, 64, 32, 4, False, True)
    fbs = rocprof_bench.symbol_to_config("void fa::fa_fwd_kernel<15, 1, 8, 64, true, true, true, false, false, false, 128, 0>(fa::KernelArgs)")
    assert fbs.optimized_softmax and not kc.wants_speculative(fbs) and kc.softmax_mode(fbs) == "first_block_skip"   # register-staged
    cfg_ks = rocprof_bench.symbol_to_config("void fa::fa_fwd_kernel<5, 1, 4, 64, true, true, false, true, true, false, 128, 0, 2>(fa::KernelArgs)")
    assert (cfg_ks.B_r, cfg_ks.B_c, cfg_ks.n_warps, cfg_ks.mma_double_buffer_loads) == (64, 64, 4, True)
    cfg_spec = rocprof_bench.symbol_to_config("void fa::fa_fwd_kernel64<15, false, 0, false, true>(fa::KernelArgs)")
    # (a kernel symbol names a device variant: the adaptive mode is a launch-time policy over two of them)
    assert cfg_spec == replace(kc.best_config(kc.DType.BF16), adaptive_softmax=False) and kc.softmax_mode(cfg_spec) == "speculative"
    cfg64 = rocprof_bench.symbol_to_config("void fa::fa_fwd_kernel64<15, false, 0>(fa::KernelArgs)")
    assert cfg64 == kc.FlashForwardKernelConfig(kc.DType.BF16, 128, 256, 64, 4, True, True, True, 0, 0, 0, True, False)
    assert kernel_resources.demangle_variant("_ZN2fa15fa_fwd_kernel64ILi5ELb1ELi0EEEvNS_10KernelArgsE")["masked"] == 2
    spec64 = kernel_resources.demangle_variant("_ZN2fa15fa_fwd_kernel64ILi15ELb1ELi0ELb1ELb1EEEvNS_10KernelArgsE")
    assert (spec64["masked"], spec64["opt_softmax"], spec64["rows_per_wave"]) == (3, 1, 64)
    ks = kernel_resources.demangle_variant("_ZN2fa13fa_fwd_kernelILi5ELi1ELi4ELi64ELb1ELb1ELb1ELb1ELb1ELb0ELi128ELi0ELi2EEEvNS_10KernelArgsE")
    assert (ks["rows_per_wave"], ks["n_waves"], ks["opt_softmax"]) == (16, 4, 1)   # key split: B_r = 64
    assert rocprof_bench.symbol_to_config("void at::native::foo<float>()") is None
    v = kernel_resources.demangle_variant("_ZN2fa13fa_fwd_kernelILi15ELi1ELi8ELi64ELb1ELb1ELb0ELb1ELb0ELb0ELi128ELi0EEEvNS_10KernelArgsE")
    assert v == dict(dtype=15, rows_per_wave=32, n_waves=8, B_c=64, swizzled=1, eager=1, opt_softmax=0, pipelined=1, dma=0,
                     masked=0, d_head=128)


def test_resource_remark_parser():
    text = "\n".join([
        "./fa_fwd_kernel.hpp:150:1: remark: Function Name: _ZN2fa13fa_fwd_kernelILi15ELi1ELi8ELi64ELb1ELb1ELb0ELb1ELb1ELb0ELi128ELi0EEEvNS_10KernelArgsE [-Rpass-analysis=kernel-resource-usage]",
        "./fa_fwd_kernel.hpp:150:1: remark:     TotalSGPRs: 58 [-Rpass-analysis=kernel-resource-usage]",
        "./fa_fwd_kernel.hpp:150:1: remark:     VGPRs: 246 [-Rpass-analysis=kernel-resource-usage]",
        "./fa_fwd_kernel.hpp:150:1: remark:     SGPRs Spill: 7 [-Rpass-analysis=kernel-resource-usage]",
        "./fa_fwd_kernel.hpp:150:1: remark:     AGPRs: 0 [-Rpass-analysis=kernel-resource-usage]",
        "./fa_fwd_kernel.hpp:150:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]",
        "./fa_fwd_kernel.hpp:150:1: remark:     Occupancy [waves/SIMD]: 2 [-Rpass-analysis=kernel-resource-usage]",
        "./fa_fwd_kernel.hpp:150:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]",
    ])
    (row,) = kernel_resources.parse_remarks(text)
    assert row["vgprs"] == 246 and row["agprs"] == 0 and row["scratch_bytes"] == 0 and row["occupancy"] == 2
    assert row["vgpr_spill"] == 0 and row["n_waves"] == 8
    assert row["sgprs"] == 58 and row["sgpr_spill"] == 7


def test_rocprof_csv_parsers(tmp_path):
    sym = "void fa::fa_fwd_kernel<15, 1, 8, 64, true, true, false, true, true, false, 128, 0>(fa::KernelArgs)"
    trace = tmp_path / "p_kernel_trace.csv"
    trace.write_text(
        "Kind,Agent_Id,Kernel_Name,Start_Timestamp,End_Timestamp,VGPR_Count,Accum_VGPR_Count,LDS_Block_Size,Scratch_Size\n"
        f'KERNEL_DISPATCH,1,"{sym}",1000,601000,248,0,65536,0\n'
        f'KERNEL_DISPATCH,1,"{sym}",2000000,2500000,248,0,65536,0\n'
        'KERNEL_DISPATCH,1,"void at::native::other()",1,2,8,0,0,0\n')
    pmc = tmp_path / "p_counter_collection.csv"
    pmc.write_text(
        "Kernel_Name,Counter_Name,Counter_Value\n"
        f'"{sym}",TCC_HIT_sum,90\n"{sym}",TCC_MISS_sum,10\n"{sym}",GRBM_GUI_ACTIVE,8000000\n')
    t = rocprof_bench.parse_kernel_trace(str(trace))
    c = rocprof_bench.parse_counters(str(pmc))
    rows = rocprof_bench.table_rows(t, c, 4, 16, 4096, 128, skip_first=1)
    assert len(rows) == 1
    r = rows[0]
    assert r["kernel"] == "(BF16, 128, 256, 64, 8): async+eager+swizzled+load_0_0_0_tiles+buffer"
    assert abs(r["dur_ms"] - 0.5) < 1e-9 and r["vgpr"] == 248 and r["lds"] == 65536
    assert abs(r["l2_hit"] - 90.0) < 1e-9 and r["cycles"] == 1e6
    assert abs(r["mfma_tflops"] - 549755813888 / 0.5e-3 / 1e12) < 1e-6


def test_generated_variant_list_is_current_and_covers_every_config():
    from flash_attention_from_scratch_amd.tools import generate_kernel_instantiations as gen
    from flash_attention_from_scratch_amd import _capi

    assert gen.main(["--check"]) == 0
    built, masked = set(), set()
    for info in _capi.kernels():
        c = info.cfg
        # the variant's OPT template flag: the reference's first-block skip or the speculative softmax (softmax_mode says which)
        opt = _capi.SOFTMAX_MODES[info.softmax_mode] in ("first_block_skip", "speculative")
        key = (c.dtype, info.rows_per_wave, c.n_warps, c.B_c, bool(c.swizzled), bool(c.eager_load_blocks),
               opt, bool(c.mma_double_buffer_loads), bool(c.async_copy), c.d_head, bool(info.prescaled_q))
        (masked if info.masked else built).add(key)
    wanted = {gen.variant_of(cfg) for cfg in kc.get_all_supported_configs()}
    assert wanted == built
    assert masked == {v for v in wanted if gen.has_masked_variant(v)} and masked


LINT_SAMPLE = textwrap.dedent("""\
    \tv_cvt_pk_bf16_f32 v20, v1, v2
    \tv_mfma_f32_32x32x16_bf16 a[0:15], v[16:19], v[20:23], a[0:15]
    \tv_mov_b32_e32 v17, v40
    \tv_add_f32_e32 v50, v51, v52
    \tv_mfma_f32_32x32x16_bf16 a[16:31], v[24:27], v[28:31], a[16:31]
    \tv_add_f32_e32 v60, v61, v62
    \tds_read_b128 v[24:27], v99
    \ts_endpgm
    """)


def test_isa_lint_flags_operand_war_and_raw(tmp_path):
    f = tmp_path / "k.s"
    f.write_text(LINT_SAMPLE)
    kinds = sorted(k for k, *_ in isa_lint64.lint(str(f), window=3, raw=2))
    # v_cvt_pk right in front of the MFMA that reads v20 (RAW); v_mov into v17 and the ds_read into
    # v[24:27] right behind the MFMAs that read them (WAR)
    assert kinds == ["RAW", "WAR", "WAR"]


def test_isa_lint_flags_accumulator_copies_inside_a_visit(tmp_path):
    """A visit is 32 MFMAs into VGPRs then 32 into AGPRs; a v_accvgpr_* behind its third MFMA is the
    compiler moving accumulators around inside the pinned stream (seen in a trace build), and an
    `s_nop n` between a producer and the MFMA counts as n + 1 issue slots."""
    visit = ["v_mfma_f32_32x32x16_bf16 v[0:15], v[100:103], a[128:131], v[0:15]"] * 32
    visit += ["v_mfma_f32_32x32x16_bf16 a[0:15], v[100:103], v[104:107], a[0:15]"] * 32
    clean = tmp_path / "clean.s"
    clean.write_text("\n".join(["v_mov_b32_e32 v100, 0", "s_nop 3"] + visit + ["s_endpgm"]) + "\n")
    assert isa_lint64.lint(str(clean), window=3, raw=3) == []
    close = tmp_path / "close.s"
    close.write_text("\n".join(["v_mov_b32_e32 v100, 0", "s_nop 0"] + visit + ["s_endpgm"]) + "\n")
    assert [k for k, *_ in isa_lint64.lint(str(close), window=3, raw=3)] == ["RAW"]
    bad = tmp_path / "bad.s"
    bad.write_text("\n".join(visit[:40] + ["v_accvgpr_read_b32 v200, a3"] + visit[40:] + ["s_endpgm"]) + "\n")
    kinds = [k for k, *_ in isa_lint64.lint(str(bad), window=3, raw=3)]
    assert [k for k in kinds if k != "MFMAD"] == ["AGPR"]
    assert "MFMAD" in kinds   # (the copy also reads a3 while the MFMAs in front of it are still producing it)


def test_isa_lint_flags_compiler_uses_of_m0(tmp_path):
    """The DMA pieces leave their LDS destination in M0 from one asm statement to a later one; hipcc does
    not preserve M0 around asm, so any M0 use of its own inside a fa_fwd_kernel64 function is a finding
    (the kernel's own statements, between the ASMSTART / ASMEND markers, are not)."""
    body = textwrap.dedent("""\
        _ZN2fa15fa_fwd_kernel64ILi15ELb0ELi0ELb0ELb0EEEvNS_10KernelArgsE:
        \t;;#ASMSTART
        \ts_mov_b32 m0, s4
        \t;;#ASMEND
        \tv_add_f32_e32 v1, v2, v3
        %s
        \t;;#ASMSTART
        \tglobal_load_lds_dwordx4 v7, s[0:1]
        \t;;#ASMEND
        \ts_endpgm
        .Lfunc_end0:
        """)
    ok = tmp_path / "ok.s"
    ok.write_text(body % "\tv_mul_f32_e32 v4, v5, v6")
    assert isa_lint64.lint(str(ok), only="fa_fwd_kernel64") == []
    for offender in ("\ts_mov_b32 m0, s9", "\ts_set_gpr_idx_on s3, gpr_idx(SRC0)", "\tv_movrels_b32_e32 v1, v2"):
        bad = tmp_path / "bad.s"
        bad.write_text(body % offender)
        kinds = [k for k, *_ in isa_lint64.lint(str(bad), only="fa_fwd_kernel64")]
        assert kinds[:1] == ["M0"] and set(kinds) <= {"M0", "M0GAP"}, offender   # (a write right in front of the DMA is also M0GAP)
    # another function of the same file is not held to the rule, and `only` skips it altogether
    other = tmp_path / "other.s"
    other.write_text(body.replace("15fa_fwd_kernel64", "13fa_fwd_kernel") % "\ts_mov_b32 m0, s9")
    assert [k for k, *_ in isa_lint64.lint(str(other))] == ["M0GAP"]   # (the missing wait state is a finding in any function)
    assert isa_lint64.lint(str(other), only="fa_fwd_kernel64") == []


def test_isa_lint_flags_writes_into_the_data_of_a_wide_store(tmp_path):
    """A store of more than 64 bits keeps reading its data registers for two wait states; hipcc pads that for its own
    stores only -- the epilogue's are asm.  (Seen: the register allocator reused a stored v[i] for the next store's
    address, one instruction behind the store; rows of O came out as garbage.)"""
    body = "\n".join(["global_store_dwordx4 v120, v[66:69], s[0:1] nt", "%s", "v_add_u32_e32 v66, 12, v98", "s_endpgm"]) + "\n"
    for filler, expect in (("", ["STDATA"]), ("s_nop 0", ["STDATA"]), ("s_nop 1", []), ("v_mov_b32_e32 v1, v2\nv_mov_b32_e32 v3, v4", [])):
        f = tmp_path / "st.s"
        f.write_text(body % filler)
        assert [k for k, *_ in isa_lint64.lint(str(f))] == expect, filler
    other = tmp_path / "other.s"   # a write to another register, a 64-bit store, a compare (writes no VGPR): no finding
    other.write_text("\n".join(["global_store_dwordx4 v120, v[66:69], s[0:1]", "v_add_u32_e32 v70, 12, v98",
                                "global_store_dwordx2 v120, v[66:67], s[0:1]", "v_add_u32_e32 v66, 12, v98",
                                "buffer_store_dwordx4 v[10:13], v1, s[4:7], 0 offen", "v_cmp_lt_f32_e32 vcc, v10, v2", "s_endpgm"]) + "\n")
    assert isa_lint64.lint(str(other)) == []
    buf = tmp_path / "buf.s"
    buf.write_text("buffer_store_dwordx4 v[10:13], v1, s[4:7], 0 offen\nv_mov_b32_e32 v12, 0\ns_endpgm\n")
    assert [k for k, *_ in isa_lint64.lint(str(buf))] == ["STDATA"]


def test_isa_lint_flags_early_use_of_an_mfma_result(tmp_path):
    """The result of an 8-pass MFMA exists passes + 3 = 11 wait states behind it (7 for the 4-pass 16x16x32); hipcc
    sees a one-cycle asm statement.  Accumulating into exactly the same registers is the legal back-to-back use;
    what stands behind an unconditional branch in the listing is another path."""
    mf = "v_mfma_f32_32x32x16_bf16 v[0:15], v[100:103], v[104:107], v[0:15]"
    def kinds(lines):
        f = tmp_path / "m.s"
        f.write_text("\n".join(lines + ["s_endpgm"]) + "\n")
        return [k for k, *_ in isa_lint64.lint(str(f), window=0, raw=0)]
    assert kinds([mf, "v_exp_f32_e32 v200, v3"]) == ["MFMAD"]                       # read too early
    assert kinds([mf, "s_nop 7", "s_nop 1", "v_exp_f32_e32 v200, v3"]) == ["MFMAD"]  # 10 wait states
    assert kinds([mf, "s_nop 7", "s_nop 2", "v_exp_f32_e32 v200, v3"]) == []        # 11
    assert kinds([mf, mf, mf]) == []                                                # same accumulator, back to back
    assert kinds([mf, "v_mfma_f32_32x32x16_bf16 v[16:31], v[0:3], v[104:107], v[16:31]"]) == ["MFMAD"]  # as an A operand
    assert kinds([mf, "v_mov_b32_e32 v5, 0"]) == ["MFMAD"]                          # overwritten before it lands
    assert kinds([mf, "s_branch .LBB0_3", "v_exp_f32_e32 v200, v3"]) == []
    m16 = "v_mfma_f32_16x16x32_bf16 a[0:3], v[100:103], v[104:107], a[0:3]"
    assert kinds([m16, "s_nop 5", "v_accvgpr_read_b32 v9, a2"]) == ["MFMAD"]
    assert kinds([m16, "s_nop 6", "v_accvgpr_read_b32 v9, a2"]) == []


def test_isa_lint_flags_permlane_swap_right_behind_its_producer(tmp_path):
    def kinds(lines):
        f = tmp_path / "p.s"
        f.write_text("\n".join(lines + ["s_endpgm"]) + "\n")
        return [k for k, *_ in isa_lint64.lint(str(f), window=0, raw=0)]
    assert kinds(["v_max_f32 v0, v1, v2", "v_permlane32_swap_b32_e32 v0, v68"]) == ["PERMSW"]
    assert kinds(["v_max_f32 v0, v1, v2", "v_mov_b32_e32 v68, v0", "v_permlane32_swap_b32_e32 v0, v68"]) == ["PERMSW"] * 2  # copy 0, producer 1 behind
    assert kinds(["v_max_f32 v0, v1, v2", "v_mov_b32_e32 v68, v0", "s_nop 1", "v_permlane32_swap_b32_e32 v0, v68"]) == []
    assert kinds(["v_max_f32 v9, v1, v2", "v_permlane16_swap_b32_e32 v0, v68"]) == []


def test_isa_lint_flags_dma_right_behind_its_m0_write(tmp_path):
    f = tmp_path / "m0.s"
    f.write_text(";;#ASMSTART\ns_mov_b32 m0, s4\nglobal_load_lds_dwordx4 v7, s[0:1]\n;;#ASMEND\ns_endpgm\n")
    assert [k for k, *_ in isa_lint64.lint(str(f), window=0, raw=0)] == ["M0GAP"]
    f.write_text(";;#ASMSTART\ns_mov_b32 m0, s4\ns_nop 0\nglobal_load_lds_dwordx4 v7, s[0:1]\n;;#ASMEND\ns_endpgm\n")
    assert isa_lint64.lint(str(f), window=0, raw=0) == []


def test_isa_lint_flags_vector_written_sgpr_read_by_a_memory_instruction(tmp_path):
    """hipcc parks scalars in VGPR lanes and reloads them with v_readlane; a vector-memory instruction may read such
    an SGPR as its base only 5 wait states later.  hipcc pads its own loads and stores, not the asm DMA pieces."""
    body = "\n".join(["v_readlane_b32 s7, v238, 44", ";;#ASMSTART", "%s", "global_load_lds_dwordx4 v6, s[6:7]", ";;#ASMEND", "s_endpgm"]) + "\n"
    cases = (("s_mov_b32 m0, s0\ns_nop 0", ["SGPRVM"]),            # 2 wait states
             ("s_mov_b32 m0, s0\ns_nop 3", []),                     # 5
             ("s_add_u32 s6, s6, s10\ns_addc_u32 s7, s7, s11", []))  # rewritten by scalar instructions: their result is read
    for filler, expect in cases:
        f = tmp_path / "sg.s"
        f.write_text(body % filler)
        assert [k for k, *_ in isa_lint64.lint(str(f))] == expect, filler
    other = tmp_path / "other.s"   # another register pair: no finding
    other.write_text("v_readfirstlane_b32 s9, v3\nglobal_store_dwordx2 v1, v[2:3], s[4:5]\ns_endpgm\n")
    assert isa_lint64.lint(str(other)) == []


def test_isa_lint_on_the_built_64_row_kernels(tmp_path):
    """Compile the two hand-placed kernels to ISA (as the library build does) and require that no
    instruction near an inline-asm MFMA touches its operand registers: hipcc cannot see these
    hazards, the schedule has to avoid them by construction."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "flash_attention_from_scratch_amd", "csrc")
    src = tmp_path / "probe.hip"
    src.write_text('#include "fa_registry.hpp"\nnamespace fa { const KernelEntry kE[] = {'
                   " make_entry<15, 2, 4, 64, true, true, false, true, true>(),"
                   " make_entry<5, 2, 4, 64, true, true, false, true, true>(),"
                   " make_entry<15, 2, 4, 64, true, true, false, true, true, true>(),"
                   " make_entry<15, 2, 4, 64, true, true, true, true, true>(),"     # opt_softmax: the speculative build
                   " make_entry<5, 2, 4, 64, true, true, true, true, true>() }; }\n")
    out = tmp_path / "probe.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-S",
                    "--cuda-device-only", "-I", csrc, str(src), "-o", str(out)], check=True, timeout=600)
    text = out.read_text()
    assert text.count("v_mfma_f32_32x32x16") >= 3 * (32 + 4 * 64) + 2 * (2 * 32 + 8 * 64)
    assert "scratch_" not in text
    assert isa_lint64.lint(str(out), window=4, raw=3) == []
    assert len(isa_lint64.split_kernels(str(out))) >= 6  # (plain, plain fp16, causal, ragged, speculative x 2)


def test_visit_histogram_matches_the_committed_digest():
    """Toolchain pin (profiles/r06/toolchain.json, tools/isa_digest.py): the hand-placed visit of the persistent kernel
    must come out of THIS hipcc as the plan dealt it -- per visit 64 MFMAs, 64 v_exp_f32, 64 v_fmamk (c applied in fp32,
    softmax.cuh:51-64), 64 row-sum adds, 32 packs, 48 LDS operand reads (16 ds_read_b128 + 32 ds_read_b64_tr_b16), 8 LDS-DMA
    pieces, and in the speculative first pass no row max -- and as the digest the round's measurements belong to recorded
    it.  Round 5: a walk is [first group | hot loop | last two groups]; hipcc merges the four HOT visits
    into one block of 256 MFMAs, which must carry nothing but the plan: no lane spill, no accumulator copy, <= 385
    instructions per visit.  A compiler upgrade (or any source change) that moves it fails here: look at the new ISA,
    re-measure, then regenerate the digest (python flash_attention_from_scratch_amd/tools/isa_digest.py --write
    profiles/r06/toolchain.json)."""
    import json

    from flash_attention_from_scratch_amd.tools import isa_digest

    got = isa_digest.digest()
    assert got is not None, "the build keeps the ISA of every slice under csrc/build (make -C flash_attention_from_scratch_amd/csrc)"
    for name, blocks in got["kernels"].items():
        assert blocks, name
        for i, v in enumerate(blocks):
            n = max(1, round(v["mfma"] / 64))          # visits hipcc merged into this block
            head_split = v["mfma"] == 64 * n - 3        # (the barrier two MFMAs into a visit split off a 3-MFMA head)
            assert v["mfma"] == 64 * n or head_split, (name, i, v)
            assert v["ds_read_b128"] == 16 * n - (2 if head_split else 0), (name, i, v)
            assert v["v_exp_f32"] in (64 * n, 64 * n + 1), (name, i, v)
            assert (v["v_cvt_pk"], v["ds_read_b64_tr_b16"], v["global_load_lds_dwordx4"]) == (32 * n, 32 * n, 8 * n), (name, i, v)
            assert v["v_writelane_b32"] <= 2, (name, i, v)
        # one walk (lazy) or two (speculative: the first pass has no row max at all, its second pass keeps the running max)
        first_pass = [v for v in blocks if v["v_max3_f32"] == 0]
        second = [v for v in blocks if v["v_max3_f32"] > 0]
        n_visits = lambda bs: sum(max(1, round(v["mfma"] / 64)) for v in bs)  # noqa: E731
        if "speculative" in name:
            # (round 6: the plain speculative kernel redoes failed items as 128-row half items, one 32-row Q tile per wave:
            # its second walk is twelve blocks of 16 + 16 MFMAs with the running max; the pre-scaled-Q build keeps the 64-row one)
            half = got["second_pass_half_visits"].get(name, [])
            if "prescaled" in name:
                assert n_visits(second) == 12 and not half, (name, n_visits(second), len(half))
            else:
                assert not second and len(half) == 12, (name, n_visits(second), len(half))
                for v in half:
                    assert v["mfma"] in (30, 32) and v["v_cvt_pk"] == 16 and v["v_fmamk_f32"] == 32, (name, v)
                    assert (v["ds_read_b64_tr_b16"], v["global_load_lds_dwordx4"]) == (32, 8), (name, v)
            assert n_visits(first_pass) == 12, (name, n_visits(first_pass))
            fmamk = 0 if "prescaled" in name else 64   # pre-scaled Q: no multiply per logit at all in the first pass
            assert all(v["v_fmamk_f32"] == fmamk * max(1, round(v["mfma"] / 64)) for v in first_pass), name
            hot = [v for v in first_pass if v["mfma"] == 256]
            assert len(hot) == 1, (name, [v["mfma"] for v in first_pass])
            h = hot[0]
            assert h["v_readlane_b32"] == 0 and h["v_writelane_b32"] == 0 and h["v_accvgpr"] == 0, (name, h)
            assert h["v_add_f32"] == 4 * 64 + 2, (name, h)   # 64 row-sum adds per visit + the guard's two per four visits
            assert h["instructions"] <= 4 * 385, (name, h)
        else:
            assert not first_pass and n_visits(second) == 12, (name, n_visits(first_pass), n_visits(second))
    want = json.load(open(os.path.join(ROOT, "profiles", "r06", "toolchain.json")))
    assert got["hipcc"] == want["hipcc"], ("the compiler changed: every hand-placed schedule needs re-verification on the GPU "
                                           "(pytest -m gpu, tools/soak.py, bench.py) before the digest is regenerated", got["hipcc"])
    assert got["kernels"] == want["kernels"], "the visit's instruction histogram moved: see the docstring"
    assert got["second_pass_half_visits"] == want["second_pass_half_visits"], "the half-item visit's histogram moved"


def test_product_translation_units_cannot_instantiate_timing_only_variants(tmp_path):
    """VERDICT r05 task 8: the persistent kernel's experiment bits (template argument ABL: most of them compute WRONG
    results on purpose, for timing) exist only in tools built with -DFA_TUNE.  The product's slices are compiled without it
    -- checked on the PREPROCESSED source of a product slice: the constant the kernel body reads is 0 there and a
    static_assert refuses any other ABL -- and the Makefile gives -DFA_TUNE to the tools alone.  A translation unit without
    the macro that asks for ABL = 2 (no softmax vector work at all) does not compile."""
    import subprocess
    csrc = os.path.join(ROOT, "flash_attention_from_scratch_amd", "csrc")
    hipcc = "/opt/rocm/bin/hipcc"
    pre = subprocess.run([hipcc, "-E", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-DFA_INST_DT=15", "-DFA_INST_QT=2",
                          os.path.join(csrc, "fa_inst.hip")], capture_output=True, text=True, timeout=600)
    assert pre.returncode == 0, pre.stderr[-2000:]
    text = pre.stdout
    assert "constexpr int TUNE = 0;" in text and "constexpr int TUNE = ABL;" not in text
    assert 'static_assert(ABL == 0' in text
    mk = open(os.path.join(csrc, "Makefile")).read()
    flags = {ln.split(":=")[0].split("?=")[0].strip(): ln for ln in mk.splitlines() if ":=" in ln or "?=" in ln}
    assert "-DFA_TUNE" in flags["TOOLFLAGS"] and "-DFA_TUNE" not in flags["HIPFLAGS"] and "-DFA_TUNE" not in flags["QT2FLAGS"]
    src = tmp_path / "wrong.hip"
    src.write_text('#include "%s/fa_fwd_kernel64.hpp"\n'
                   'template __global__ void fa::fa_fwd_kernel64<15, false, 2, false, true, false, 2>(const fa::KernelArgs);\n' % csrc)
    bad = subprocess.run([hipcc, "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-fsyntax-only", str(src)],
                         capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "timing-only variants" in bad.stderr, bad.stderr[-1500:]
    # (the header the review reads: the plans and the trace helpers live in their own files)
    n_lines = len(open(os.path.join(csrc, "fa_fwd_kernel64.hpp")).read().splitlines())
    assert n_lines <= 1400, n_lines
