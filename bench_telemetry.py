"""Box telemetry for bench.py: shader / memory clock, socket power and temperature of the GPU the bench runs on, so that a line can be
read as "this box ran at X MHz under Y W" (nothing in the line told a throttled box from a regression).

Two sources, best effort, never fatal:
  * sysfs (amdgpu hwmon: `freq1_input`, `power1_average` | `power1_input`, `temp*_input`; `pp_dpm_sclk` / `pp_dpm_mclk`) -- plain file reads,
    cheap enough for a sampling thread DURING the timed windows;
  * `amd-smi metric --json` / `rocm-smi --json` -- one subprocess started beside a timed window (it runs on another host core) and collected
    afterwards; its JSON is searched for the clock / power / temperature leaves by key pattern, since the schema differs between releases.
Every reading is a flat dict of scalars (what the driver's BENCH record keeps).
"""
import glob
import json
import os
import re
import shutil
import subprocess
import threading
import time


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _cards():
    out = []
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if (_read(os.path.join(dev, "vendor")) or "").lower() == "0x1002":
            out.append(dev)
    return out


def pci_address(device_index):
    """'0000:72:00.0' of HIP device `device_index` in this process (hipDeviceGetPCIBusId of the runtime torch has loaded; falls back to
    the integer fields of torch's device properties), or None."""
    import ctypes
    try:
        # the copy of the HIP runtime ALREADY mapped into this process (torch's): opening it by its path returns that very handle,
        # opening it by bare name could map a second runtime from /opt/rocm
        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        if path:
            hip = ctypes.CDLL(path)
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(device_index)) == 0 and buf.value:
                return buf.value.decode().lower()
    except (OSError, AttributeError):
        pass
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        return "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
    except Exception:                                        # noqa: BLE001  (telemetry is best effort)
        return None


def find_card(pci_bus_id=None):
    """sysfs device directory of the GPU with this PCI address ('0000:05:00.0'), else of the only / first amdgpu card, else None."""
    cards = _cards()
    if isinstance(pci_bus_id, str) and pci_bus_id:
        want = pci_bus_id.lower()
        for dev in cards:
            if os.path.realpath(dev).lower().endswith(want):
                return dev
    return cards[0] if cards else None


def _dpm_current(text):
    """'0: 132Mhz\\n1: 2100Mhz *' -> 2100.0 (the line marked with *)."""
    if not text:
        return None
    for line in text.splitlines():
        if "*" in line:
            m = re.search(r"(\d+(?:\.\d+)?)\s*mhz", line, re.I)
            if m:
                return float(m.group(1))
    return None


def sysfs_snapshot(dev):
    """One reading from amdgpu's sysfs files under `dev` (a /sys/class/drm/cardN/device directory)."""
    if not dev:
        return {}
    out = {}
    for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
        v = _read(os.path.join(hw, "freq1_input"))
        if v and v.isdigit():
            out["sclk_MHz"] = round(int(v) / 1e6, 1)
        v = _read(os.path.join(hw, "freq2_input"))
        if v and v.isdigit():
            out["mclk_MHz"] = round(int(v) / 1e6, 1)
        for name in ("power1_average", "power1_input"):
            v = _read(os.path.join(hw, name))
            if v and v.isdigit():
                out["power_W"] = round(int(v) / 1e6, 1)
                break
        v = _read(os.path.join(hw, "power1_cap"))
        if v and v.isdigit():
            out["power_cap_W"] = round(int(v) / 1e6, 1)
        temps = []
        for t in glob.glob(os.path.join(hw, "temp*_input")):
            v = _read(t)
            if v and v.lstrip("-").isdigit():
                temps.append(int(v) / 1e3)
        if temps:
            out["temp_C"] = round(max(temps), 1)
    if "sclk_MHz" not in out:
        v = _dpm_current(_read(os.path.join(dev, "pp_dpm_sclk")))
        if v is not None:
            out["sclk_MHz"] = v
    if "mclk_MHz" not in out:
        v = _dpm_current(_read(os.path.join(dev, "pp_dpm_mclk")))
        if v is not None:
            out["mclk_MHz"] = v
    v = _read(os.path.join(dev, "gpu_busy_percent"))
    if v and v.isdigit():
        out["busy_pct"] = int(v)
    if out:
        out["source"] = "sysfs"
    return out


def _leaves(obj, path=""):
    if isinstance(obj, dict):
        if "value" in obj and not isinstance(obj["value"], (dict, list)):
            yield path, obj["value"]
            return
        for k, v in obj.items():
            yield from _leaves(v, path + "." + str(k).lower())
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            yield from _leaves(v, path + "[%d]" % i)
    else:
        yield path, obj


def _num(v):
    if isinstance(v, (int, float)):
        return float(v)
    if isinstance(v, str):
        m = re.match(r"\s*(-?\d+(?:\.\d+)?)", v)
        if m:
            return float(m.group(1))
    return None


PATTERNS = (("sclk_MHz", (r"gfx.*\.clk$", r"gfx_0.*clk", r"gfx.*cur", r"sclk", r"current_gfxclk")),
            ("mclk_MHz", (r"mem.*\.clk$", r"mem_0.*clk", r"mclk", r"current_uclk")),
            ("power_W", (r"socket_power", r"current_socket_power", r"average_socket_power", r"power.*avg", r"\.power$")),
            ("temp_C", (r"hotspot", r"junction", r"temperature.*edge", r"temp")))


def parse_smi_json(text):
    """Pick clock / power / temperature leaves out of an amd-smi / rocm-smi JSON dump by key pattern (first GPU)."""
    try:
        data = json.loads(text)
    except ValueError:
        return {}
    leaves = list(_leaves(data))
    out = {}
    for key, pats in PATTERNS:
        for pat in pats:
            hit = next(((p, _num(v)) for p, v in leaves if re.search(pat, p) and _num(v) is not None and "limit" not in p
                        and "max" not in p and "min" not in p), None)
            if hit:
                out[key] = round(hit[1], 1)
                break
    return out


def smi_command():
    if shutil.which("amd-smi"):
        return ["amd-smi", "metric", "--json"]
    if shutil.which("rocm-smi"):
        return ["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"]
    return None


def smi_snapshot(timeout=8.0):
    cmd = smi_command()
    if not cmd:
        return {}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    except (OSError, subprocess.SubprocessError):
        return {}
    out = parse_smi_json(r.stdout)
    if out:
        out["source"] = cmd[0]
    return out


def snapshot(dev=None):
    """sysfs when it answers, else one SMI call."""
    s = sysfs_snapshot(dev)
    return s if s else smi_snapshot()


class LoadSampler:
    """Readings WHILE the GPU is under the bench's load: a sysfs polling thread (every `period` s), or -- without sysfs -- one SMI
    subprocess started by start() and collected by stop().  stop() returns flat scalars: sclk_MHz_{min,mean,max}, power_W_{mean,max}, ..."""

    def __init__(self, dev=None, period=0.02):
        self.dev, self.period = dev, period
        self.rows, self._stop, self._thr, self._proc = [], threading.Event(), None, None
        self.mode = "sysfs" if sysfs_snapshot(dev) else ("smi" if smi_command() else "none")

    def _loop(self):
        while not self._stop.is_set():
            s = sysfs_snapshot(self.dev)
            if s:
                self.rows.append(s)
            time.sleep(self.period)

    def start(self):
        if self.mode == "sysfs":
            self._thr = threading.Thread(target=self._loop, daemon=True)
            self._thr.start()
        elif self.mode == "smi":
            try:
                self._proc = subprocess.Popen(smi_command(), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            except OSError:
                self._proc = None

    def stop(self):
        out = {"source": self.mode}
        if self._thr is not None:
            self._stop.set()
            self._thr.join(timeout=2.0)
            for key in ("sclk_MHz", "mclk_MHz", "power_W", "temp_C"):
                vals = [r[key] for r in self.rows if key in r]
                if vals:
                    out[key + "_min"], out[key + "_mean"], out[key + "_max"] = min(vals), round(sum(vals) / len(vals), 1), max(vals)
            out["samples"] = len(self.rows)
        elif self._proc is not None:
            try:
                text, _ = self._proc.communicate(timeout=10.0)
                out.update(parse_smi_json(text))
            except (OSError, subprocess.SubprocessError):
                self._proc.kill()
        return out
