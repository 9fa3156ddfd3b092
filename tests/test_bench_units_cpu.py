"""CPU tests of the bench line's building blocks that need no GPU: the CPU leg (thread probe, sub-batches, per-block host rates), the
telemetry readers (amdgpu sysfs layout, amd-smi / rocm-smi JSON by key pattern, card matched by PCI address) and the launch flags."""
import json
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bench_telemetry as tele  # noqa: E402


KEYS14 = ["SE", "CBAM", "ECA", "ViTAttn", "CSWin_s1", "CSWin_s2", "CSWin_s3", "CSWin_s4", "XCABlock", "XCA", "DA64", "DA256", "Mixer", "ViTBase"]


def test_cpu_leg_reports_flat_scalars_per_block():
    torch.manual_seed(0)
    w = torch.randn(64, 64)
    blocks = [dict(name="toy GEMM", key="Toy", x=torch.randn(32, 8, 64), bound="mfma", work=2.0 * 8 * 64 * 64 * 32, cpu=lambda xs: xs @ w, cpu_n=16),
              dict(name="toy scale", key="Scale", x=torch.randn(32, 4, 8, 8), bound="hbm", work=2.0 * 4 * 8 * 8 * 4 * 32, cpu=lambda xs: xs * 2.0, cpu_n=64)]
    args = types.SimpleNamespace(cpu_sample=None)
    flat, detail = bench.cpu_baseline(blocks, args)
    assert flat["kind"] == "port" and flat["value"] > 0 and flat["cores"] >= 1
    for k in ("img_s_Toy", "img_s_Scale"):
        assert k in flat and isinstance(flat[k], (int, float)), k
    assert "Toy=" in flat["threads_x_sub"] and "Toy=" in flat["GFLOPs"] and "Scale=" in flat["GBps"]
    assert all(not isinstance(v, (list, dict)) for v in flat.values()), "the driver's record keeps scalar leaves only"
    assert [d["key"] for d in detail] == ["Toy", "Scale"] and detail[0]["images"] == 16 and detail[1]["images"] == 32
    assert len(detail[0]["probe_s"]) >= 1 and len(detail[0]["passes_s"]) >= 3, "best pair timed >= 3 passes (median)"
    # truthful cores: what this process may use, not what /proc/cpuinfo lists
    assert flat["affinity_cpus"] == len(os.sched_getaffinity(0)) and flat["cores"] <= flat["affinity_cpus"]
    assert "affinity %d" % flat["affinity_cpus"] in flat["host"]
    assert len(json.dumps(flat)) < 2500


def test_cpu_leg_fits_the_drivers_key_cap_with_14_blocks():
    """14 blocks: value / unit / cores / kind / sample / host + 14 img_s_* + threads_x_sub / GFLOPs / GBps / legend = 24 keys exactly."""
    blocks = [dict(name=k, key=k, x=torch.randn(16, 4, 8), bound="mfma" if i % 2 else "hbm", work=1e6, cpu=lambda xs: xs + 1.0, cpu_n=16)
              for i, k in enumerate(KEYS14)]
    os.environ["MI355_CPU_BLOCK_BUDGET_S"] = "0.05"
    try:
        flat, _ = bench.cpu_baseline(blocks, types.SimpleNamespace(cpu_sample=None))
    finally:
        del os.environ["MI355_CPU_BLOCK_BUDGET_S"]
    kept = list(flat)[:bench.DRIVER_KEYS_PER_DICT]
    for k in ("value", "unit", "cores", "kind", "sample", "host", "threads_x_sub", "GFLOPs", "GBps", "legend"):
        assert k in kept, k
    for k in KEYS14:
        assert "img_s_" + k in kept, k


def _fake_measurement():
    per_block = []
    for i, k in enumerate(KEYS14):
        hbm = k in ("SE", "CBAM", "ECA", "DA64", "DA256")
        rec = dict(block=k + "(...)", key=k, ms=0.3 + 0.01 * i, bound="hbm" if hbm else "mfma", achieved=500.0 + i, unit="GB/s" if hbm else "TFLOP/s",
                   frac=0.1 + 0.01 * i, traffic=2_000_000_000 + i)
        rec["alg_bytes"] = 1_000_000_000
        rec["hbm_frac_alg"], rec["hbm_frac_pmc"], rec["traffic_x"] = 0.2, 0.4, 2.0
        if not hbm or k.startswith("DA"):
            rec["strict_ms"] = 3.0 * rec["ms"]
        per_block.append(rec)
    calib = {"before": {"stream_copy_GBps": 4800.0, "mfma_16x16x32_TFLOPs": 1900.0, "mfma_32x32x16_TFLOPs": 1750.0, "sclk_MHz_counter": 1900.0,
                        "sclk_MHz_issue": 1700.0},
             "after": {"stream_copy_GBps": 4700.0, "mfma_16x16x32_TFLOPs": 1950.0, "mfma_32x32x16_TFLOPs": 1760.0, "sclk_MHz_counter": 1910.0,
                       "sclk_MHz_issue": 1690.0},
             "idle": {"sclk_MHz": 2400.0, "power_W": 300.0, "source": "sysfs"},
             "load": {"sclk_MHz_mean": 2050.0, "power_W_mean": 1280.0, "samples": 30, "source": "sysfs"}}
    dominant = {"name": "gemm16_pa_kernel<f16,out32>", "avg_us": 180.0, "achieved": 800.0, "frac": 0.32, "share_of_block": 0.37,
                "launches_per_forward": 24.0, "us_per_forward": 4300.0, "traced_us_per_forward": 11700.0}
    return dict(batch=256, steps=20, warmup=5, world=1, value=15800.0, ms_per_step=16.2, extra_ms=[16.21, 16.22], workload="north-star step",
                dtype="f32/f16", precision=1, dist_backend="none", gather="none (single rank)", rccl_self_test="not run (single rank)",
                ranks_seen=1, distinct_gpus=1, per_block=per_block, calib=calib, pmc_note="PMC note", dominant=dominant, dom=per_block[-1],
                blocks=None, model_ms={"CSWinT": 7.02, "XCiTnano": 2.5, "Mixer12": 4.95})


def test_line_survives_the_drivers_24_key_cap():
    """BENCH_r05.parsed lost 8 of 14 blocks and the whole box calibration because the driver keeps the first 24
    keys of each dict.  With the round-6 order the first 24 keys of `config` alone carry every block's ms and the yardstick, and the
    first 24 of `roofline` every block's fraction (one string per series)."""
    out = bench.assemble_line(_fake_measurement())
    cap = bench.DRIVER_KEYS_PER_DICT
    cfg = list(out["config"])[:cap]
    assert cfg[:4] == ["workload", "precision", "ranks_seen", "gather"]
    for k in ("stream_copy_GBps", "mfma_16x16x32_TFLOPs", "mfma_32x32x16_TFLOPs", "sclk_MHz_load", "power_W_load", "ms_windows"):
        assert k in cfg, k
    for k in KEYS14:
        assert "ms_" + k in cfg, k
    roof = list(out["roofline"])[:cap]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "block", "kernel", "ms", "kernel_avg_us", "kernel_frac", "fracs",
              "hbm_fracs", "traffic_x", "strict_ms", "strict_over_fast", "model_ms"):
        assert k in roof, k
    assert len(out["roofline"]) <= cap, "every roofline key survives the cap"
    r = out["roofline"]
    for k in KEYS14:
        assert k + "=" in r["fracs"], k
    assert "ViTBase=0.2/0.4" in r["hbm_fracs"] and "SE=" not in r["hbm_fracs"]        # HBM-graded blocks are in `fracs` already
    assert "CSWin_s1=3.00" in r["strict_over_fast"]
    for d in (out["config"], out["roofline"]):
        assert all(not isinstance(v, (list, dict)) for v in d.values()), "scalar leaves only"
    assert out["n_gpus"] == 1 and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert len(json.dumps(out)) < 8000


def test_shared_gpu_run_is_not_reported_as_a_scaling_point():
    """two gloo ranks on one GPU must not read as n_gpus = 2 / weak scaling in a record that keeps scalars only."""
    m = _fake_measurement()
    m.update(world=2, distinct_gpus=1, ranks_seen=2, dist_backend="gloo")
    out = bench.assemble_line(m)
    assert out["n_gpus"] == 1 and out["scaling"].startswith("none") and out["config"]["ranks_seen"] == 2
    assert out["config"]["distinct_gpus"] == 1


def test_sysfs_snapshot_and_card_matching(tmp_path, monkeypatch):
    root = tmp_path / "fake_drm"
    for i, pci in enumerate(("0000:05:00.0", "0000:72:00.0")):
        real = tmp_path / "devices" / pci
        hw = real / "hwmon" / "hwmon3"
        hw.mkdir(parents=True)
        (real / "vendor").write_text("0x1002\n")
        (hw / "freq1_input").write_text(str((1500 + 500 * i) * 1000000))
        (hw / "freq2_input").write_text("2000000000")
        (hw / "power1_input").write_text(str((300 + 1000 * i) * 1000000))
        (hw / "power1_cap").write_text("1400000000")
        (hw / "temp2_input").write_text("51000")
        (real / "pp_dpm_sclk").write_text("0: 500Mhz\n1: 2400Mhz *\n")
        card = root / ("card%d" % (i * 16))
        card.mkdir(parents=True)
        os.symlink(real, card / "device")
    (root / "card1").mkdir()                                    # a partition node without a device directory: ignored
    monkeypatch.setattr(tele.glob, "glob", lambda pat, _g=tele.glob.glob: _g(pat.replace("/sys/class/drm", str(root))))
    dev = tele.find_card("0000:72:00.0")
    assert dev and os.path.realpath(dev).endswith("0000:72:00.0")
    snap = tele.sysfs_snapshot(dev)
    assert snap["sclk_MHz"] == 2000.0 and snap["mclk_MHz"] == 2000.0 and snap["power_W"] == 1300.0 and snap["power_cap_W"] == 1400.0
    assert snap["temp_C"] == 51.0 and snap["source"] == "sysfs"
    assert os.path.realpath(tele.find_card(None)).endswith("0000:05:00.0")          # no address known: the first card
    assert tele.sysfs_snapshot(None) == {}
    s = tele.LoadSampler(dev, period=0.005)
    assert s.mode == "sysfs"
    s.start()
    import time
    time.sleep(0.05)
    out = s.stop()
    assert out["samples"] >= 2 and out["sclk_MHz_mean"] == 2000.0 and out["power_W_max"] == 1300.0


def test_smi_json_is_searched_by_key_pattern():
    amd = json.dumps([{"gpu": 0, "clock": {"gfx_0": {"clk": {"value": 2100, "unit": "MHz"}, "max_clk": {"value": 2400, "unit": "MHz"}},
                                           "mem_0": {"clk": {"value": 2000, "unit": "MHz"}}},
                       "power": {"socket_power": {"value": 1301, "unit": "W"}},
                       "temperature": {"hotspot": {"value": 63, "unit": "C"}, "edge": "N/A"}}])
    got = tele.parse_smi_json(amd)
    assert got == {"sclk_MHz": 2100.0, "mclk_MHz": 2000.0, "power_W": 1301.0, "temp_C": 63.0}
    rocm = json.dumps({"card0": {"sclk clock speed:": "(2100Mhz)", "Current Socket Graphics Package Power (W)": "1290.0",
                                 "Temperature (Sensor junction) (C)": "61.0"}})
    got = tele.parse_smi_json(rocm)
    assert got.get("temp_C") == 61.0
    assert tele.parse_smi_json("not json") == {}


def test_new_flags_parse():
    a = bench.parse_args(["--gpus", "2", "--dist-backend", "gloo", "--detail", "/tmp/x.json", "--no-calib"])
    assert a.dist_backend == "gloo" and a.detail == "/tmp/x.json" and a.no_calib and a.gpus == 2
    assert bench.parse_args([]).dist_backend == "nccl"
