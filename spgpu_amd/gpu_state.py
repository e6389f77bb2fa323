"""What the card was doing while a kernel was timed -- bench / experiment plumbing, no part of the product.

A memory-bound kernel on this pool runs in two or three speeds a few per cent apart: the same kernel,
the same arrays, the same box, minutes apart.  To make a number attributable the bench records next to it what can be read
without privileges: the shader / memory / fabric clocks and the socket power sampled WHILE the timed loop runs (sysfs: the
amdgpu hwmon files; a read costs microseconds, so a sampler thread does not disturb the stream), the card's identity, its
partition modes and its memory use.  Everything is best effort: a missing file is simply absent from the record.
"""
import glob
import os
import threading
import time


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def find_card(index=0):
    """sysfs device directory of the index-th AMD GPU that exposes gfx clocks, or None."""
    cards = []
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        if _read(os.path.join(dev, "vendor")) == "0x1002" and os.path.exists(os.path.join(dev, "pp_dpm_sclk")):
            cards.append(dev)
    return cards[index] if index < len(cards) else None


def _current_level(text):
    """'0: 500Mhz\\n1: 2400Mhz *' -> 2400 (the starred level), None if nothing is starred"""
    if not text:
        return None
    for line in text.splitlines():
        if line.rstrip().endswith("*"):
            digits = "".join(ch for ch in line.split(":", 1)[-1] if ch.isdigit())
            return int(digits) if digits else None
    return None


class Card:
    def __init__(self, index=0):
        self.dev = find_card(index)
        self.hwmon = None
        if self.dev:
            mons = sorted(glob.glob(os.path.join(self.dev, "hwmon", "hwmon*")))
            self.hwmon = mons[0] if mons else None

    def identity(self):
        if not self.dev:
            return {}
        out = {}
        for key, name in (("unique_id", "unique_id"), ("vbios", "vbios_version"), ("compute_partition", "current_compute_partition"),
                          ("memory_partition", "current_memory_partition"), ("perf_level", "power_dpm_force_performance_level"),
                          ("pcie", "current_link_speed")):
            v = _read(os.path.join(self.dev, name))
            if v is not None:
                out[key] = v
        used, total = _read(os.path.join(self.dev, "mem_info_vram_used")), _read(os.path.join(self.dev, "mem_info_vram_total"))
        if used and total:
            out["vram_used_gib"] = round(int(used) / 2 ** 30, 2)
            out["vram_total_gib"] = round(int(total) / 2 ** 30, 2)
        if self.hwmon:
            cap = _read(os.path.join(self.hwmon, "power1_cap"))
            if cap:
                out["power_cap_w"] = round(int(cap) / 1e6, 1)
        for clk in ("sclk", "mclk", "fclk", "socclk"):
            text = _read(os.path.join(self.dev, f"pp_dpm_{clk}"))
            if text:
                out[f"{clk}_levels"] = " ".join(text.split())
        return out

    def sample(self):
        """One reading: MHz, W, degrees C (whatever the card exposes)."""
        s = {}
        if not self.dev:
            return s
        if self.hwmon:
            for key, name, scale in (("sclk_mhz", "freq1_input", 1e-6), ("mclk_mhz", "freq2_input", 1e-6), ("power_w", "power1_average", 1e-6),
                                     ("power_w", "power1_input", 1e-6), ("temp_c", "temp1_input", 1e-3), ("temp_mem_c", "temp3_input", 1e-3)):
                if key in s:
                    continue
                v = _read(os.path.join(self.hwmon, name))
                if v and v.lstrip("-").isdigit():
                    s[key] = round(int(v) * scale, 1)
        for clk in ("sclk", "mclk", "fclk"):
            if f"{clk}_mhz" not in s:
                v = _current_level(_read(os.path.join(self.dev, f"pp_dpm_{clk}")))
                if v is not None:
                    s[f"{clk}_mhz"] = v
        for key, name in (("gpu_busy", "gpu_busy_percent"), ("mem_busy", "mem_busy_percent")):
            v = _read(os.path.join(self.dev, name))
            if v and v.isdigit():
                s[key] = int(v)
        return s


class Sampler:
    """with Sampler(card) as s: ...timed loop...; s.summary() -> {'sclk_mhz': {'min':..,'median':..,'max':..}, 'samples': n}"""

    def __init__(self, card, period_s=0.004):
        self.card, self.period = card, period_s
        self.rows, self._stop, self._thread = [], threading.Event(), None

    def __enter__(self):
        self.rows = []
        self._stop.clear()
        if self.card and self.card.dev:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def _run(self):
        while not self._stop.is_set():
            row = self.card.sample()
            row["t"] = time.perf_counter()
            self.rows.append(row)
            self._stop.wait(self.period)

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread:
            self._thread.join()
        return False

    def summary(self):
        out = {"samples": len(self.rows)}
        keys = sorted({k for r in self.rows for k in r if k != "t"})
        for k in keys:
            vals = sorted(r[k] for r in self.rows if k in r)
            if vals:
                out[k] = {"min": vals[0], "median": vals[len(vals) // 2], "max": vals[-1]}
        return out
