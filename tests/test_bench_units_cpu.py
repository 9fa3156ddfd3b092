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


def test_cpu_leg_reports_flat_scalars_per_block():
    torch.manual_seed(0)
    w = torch.randn(64, 64)
    blocks = [dict(name="toy GEMM", key="Toy", x=torch.randn(32, 8, 64), bound="mfma", work=2.0 * 8 * 64 * 64 * 32, cpu=lambda xs: xs @ w, cpu_n=16),
              dict(name="toy scale", key="Scale", x=torch.randn(32, 4, 8, 8), bound="hbm", work=2.0 * 4 * 8 * 8 * 4 * 32, cpu=lambda xs: xs * 2.0, cpu_n=64)]
    args = types.SimpleNamespace(cpu_sample=None)
    flat, detail = bench.cpu_baseline(blocks, args)
    assert flat["kind"] == "port" and flat["value"] > 0 and flat["cores"] >= 1
    for k in ("img_s_Toy", "thr_Toy", "GFLOPs_Toy", "img_s_Scale", "thr_Scale", "GBps_Scale"):
        assert k in flat and isinstance(flat[k], (int, float)), k
    assert all(not isinstance(v, (list, dict)) for v in flat.values()), "the driver's record keeps scalar leaves only"
    assert [d["key"] for d in detail] == ["Toy", "Scale"] and detail[0]["images"] == 16 and detail[1]["images"] == 32
    assert set(detail[0]["probe_s"]) <= {"8", "16", "32", "64", "128"} and len(detail[0]["probe_s"]) >= 1
    assert len(json.dumps(flat)) < 2500


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
