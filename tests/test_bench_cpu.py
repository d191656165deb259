"""CPU tier: the host-side helpers of bench.py that a GPU run relies on (statistics of the per-launch
events, the hwmon sampler, the fallback traffic records, the FLOP model)."""
import json
import os
import time

import bench
from tests.conftest import ROOT
from flash_helpers import kernel_configs as kc


def test_per_launch_distribution_fields():
    d = bench.distribution([0.5, 0.4, 0.6, 0.5])
    assert d["n"] == 4 and d["min"] == 0.4 and d["max"] == 0.6 and d["median"] == 0.5
    assert abs(d["mean"] - 0.5) < 1e-12 and abs(d["stddev"] - 0.0816496580927726) < 1e-9
    assert bench.distribution([1.0])["stddev"] == 0.0


def test_clock_sampler_reads_hwmon_files(tmp_path):
    (tmp_path / "freq1_input").write_text("1750000000\n")   # Hz
    (tmp_path / "freq2_input").write_text("2000000000\n")
    (tmp_path / "power1_input").write_text("1320000000\n")  # uW
    (tmp_path / "power1_cap").write_text("1400000000\n")
    s = bench.ClockSampler(str(tmp_path), period=0.001)
    s.start()
    time.sleep(0.02)
    s.stop()
    out = s.summary()
    assert out["sclk_mhz"]["mean"] == 1750.0 and out["mclk_mhz"]["max"] == 2000.0
    assert out["power_w"]["min"] == 1320.0 and out["power_cap_w"] == 1400.0 and out["sclk_mhz"]["n"] >= 2
    # no hwmon directory (this container, or a driver without the files): an empty summary, no error
    none = bench.ClockSampler(None)
    none.start()
    none.stop()
    assert "sclk_mhz" not in none.summary()
    missing = bench.ClockSampler(str(tmp_path / "nope"))
    missing.start()
    missing.stop()
    assert "power_w" not in missing.summary()


def test_fallback_traffic_records_match_the_default_kernels():
    """profiles/traffic_<workload>.json is what bench.py reports when rocprofv3 cannot run: it must name
    the kernel best_config() selects today, and its bytes must sit at or above the algorithmic ones."""
    for workload, (dtype_name, batch, heads, seq, d) in bench.WORKLOADS.items():
        cfg = kc.best_config(kc.DType.BF16 if dtype_name == "bf16" else kc.DType.FP16, seq)
        got = bench.committed_traffic(cfg.short_form(), workload)
        assert got is not None, workload
        algorithmic = 4 * batch * seq * heads * d * 2
        assert algorithmic <= got <= 1.6 * algorithmic, (workload, got / algorithmic)
        assert bench.committed_traffic("(BF16, 128, 64, 64, 4): something else", workload) is None
        rec = json.load(open(os.path.join(bench.ROOT, "profiles", f"traffic_{workload}.json")))
        assert rec["algorithmic_bytes"] == algorithmic


def test_flop_model_and_workload_table():
    assert bench.mfma_flop(4, 16, 4096, 128) == 549755813888           # C1, SURVEY.md 8d
    assert bench.mfma_flop(2, 32, 16384, 128) == 8796093022208         # C3
    assert bench.mfma_flop(8, 32, 8192, 128) == 8796093022208          # one GPU's shard of C4
    assert [s for s, _ in bench.C2_SWEEP] == [512, 1024, 2048, 4096, 8192, 16384]
    from flash_helpers.test.utils import BATCH_SIZE_FOR_SEQ_LEN
    assert all(BATCH_SIZE_FOR_SEQ_LEN[s] == b for s, b in bench.C2_SWEEP)


def test_wave_cycles_per_mfma_does_not_creep_up_between_rounds():
    """round 3's guard cost 0.8 wave cycles per MFMA and nothing took it back.  profiles/rNN/bench_c1.json is
    the driver command's line of a round's evidence lease; its pipe counters (SQ_WAVE_CYCLES x 4 / SQ_INSTS_MFMA over the
    whole launch) are the box-independent tracker of the kernel's instruction stream.  From round 4 on a round may not
    commit a figure more than 0.5 % above the round before it (round 3 vs round 2 is the regression this test is for)."""
    import glob
    import json
    import re

    recs = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", "bench_c1.json"))):
        rnd = int(re.search(r"r(\d\d)", path).group(1))
        try:
            line = json.load(open(path))
        except ValueError:
            continue
        pc = (line.get("roofline") or {}).get("pipe_counters") or {}
        w = line.get("wave_cycles_per_mfma", pc.get("wave_cycles_per_mfma") if isinstance(pc, dict) else None)
        if w:
            recs.append((rnd, float(w)))
    assert len(recs) >= 2 and recs[0][0] <= 2, recs
    for (r0, w0), (r1, w1) in zip(recs, recs[1:]):
        if r1 >= 4:
            assert w1 <= w0 * 1.005, f"round {r1}: {w1:.2f} wave cycles per MFMA, round {r0} had {w0:.2f}"


def test_short_timed_regions_also_report_a_sustained_one():
    """bench.py: a K-step region under 100 ms (the driver's --steps 20 at C1 is ~9 ms) also times >= 250 ms of the same
    steps with the same brackets and reports it as `sustained`; a region of 100 ms or more does not."""
    calls = []

    def step():
        calls.append(1)

    got = bench.sustained_region(step, lambda: None, lambda: None, seconds=0.010, steps=20, world=1, device=None)
    assert got["steps"] == int(0.25 / (0.010 / 20)) + 1 == len(calls) and got["seconds"] > 0 and "ms_per_step" in got
    assert bench.sustained_region(step, lambda: None, lambda: None, seconds=0.2, steps=20, world=1, device=None) is None
