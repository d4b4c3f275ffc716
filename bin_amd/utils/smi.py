"""Shader clock / package power of one GPU sampled on a host thread while a timed region runs (bench.py).

The fp32-class forward sits on the package power limit (profiles/r02_power_cap.md), so a time without the clock it was
measured at says little: `Sampler` polls the device every `period` seconds — through the in-process `amdsmi` binding
when it imports, else by parsing `rocm-smi --showclocks --showpower` — and `summary()` returns mean / min / max of both
series over the samples taken between start() and stop().  Purely observational: no device work, no effect on results.
"""
import re
import subprocess
import threading
import time


class _AmdSmi:
    name = "amdsmi"

    def __init__(self, index):
        import amdsmi
        self.m = amdsmi
        amdsmi.amdsmi_init()
        hs = amdsmi.amdsmi_get_processor_handles()
        self.h = hs[index if index < len(hs) else 0]
        self.read()                                         # fail here, not on the sampling thread

    @staticmethod
    def _num(d, keys):
        for k in keys:
            v = d.get(k) if isinstance(d, dict) else None
            if isinstance(v, (int, float)) and v == v and v > 0:
                return float(v)
        return None

    def read(self):
        m = self.m
        p = m.amdsmi_get_power_info(self.h)
        power = self._num(p, ("current_socket_power", "average_socket_power", "socket_power"))
        c = m.amdsmi_get_clock_info(self.h, m.AmdSmiClkType.GFX)
        clock = self._num(c, ("clk", "cur_clk", "current_clk"))
        if power is None and clock is None:
            raise RuntimeError(f"amdsmi returned no usable fields: {p} {c}")
        return clock, power

    def cap(self):
        try:
            d = self.m.amdsmi_get_power_cap_info(self.h)
            v = self._num(d, ("power_cap", "default_power_cap"))
            return None if v is None else (v / 1e6 if v > 1e5 else v)       # microwatts on most versions
        except Exception:
            return None


class _RocmSmi:
    name = "rocm-smi"

    def __init__(self, index):
        self.index = index
        self.read()

    def read(self):
        out = subprocess.run(["rocm-smi", "-d", str(self.index), "--showclocks", "--showpower"], capture_output=True,
                             text=True, timeout=10).stdout
        c = re.search(r"sclk clock level:\s*\d+:?\s*\((\d+)Mhz\)", out)
        p = re.search(r"(?:Current Socket|Average) Graphics Package Power \(W\):\s*([\d.]+)", out)
        if not c and not p:
            raise RuntimeError("rocm-smi output not understood")
        return (float(c.group(1)) if c else None), (float(p.group(1)) if p else None)

    def cap(self):
        return None


class Sampler:
    def __init__(self, index=0, period=0.02):
        self.period = period
        self.src, self.error = None, None
        for cls in (_AmdSmi, _RocmSmi):
            try:
                self.src = cls(index)
                break
            except Exception as e:                          # no GPU / no permission / unknown fields: report, never raise
                self.error = f"{cls.name}: {type(e).__name__}: {e}"[:200]
        self._stop = threading.Event()
        self._thread = None
        self.samples = []

    def start(self):
        self.samples = []
        if self.src is None:
            return self
        self._stop.clear()

        def loop():
            while not self._stop.is_set():
                try:
                    self.samples.append(self.src.read())
                except Exception:
                    pass
                self._stop.wait(self.period)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=5)
            self._thread = None
        return self.summary()

    def summary(self):
        if self.src is None:
            return {"source": None, "samples": 0, "error": self.error}

        def agg(vals):
            vals = [v for v in vals if v is not None]
            if not vals:
                return None
            return {"mean": round(sum(vals) / len(vals), 1), "min": round(min(vals), 1), "max": round(max(vals), 1)}
        return {"source": self.src.name, "samples": len(self.samples), "period_s": self.period,
                "clock_mhz": agg([s[0] for s in self.samples]), "power_w": agg([s[1] for s in self.samples]),
                "power_cap_w": self.src.cap()}


def sampled(fn, index=0, period=0.02, min_seconds=0.0):
    """Run fn() while sampling; returns (fn's result, summary)."""
    s = Sampler(index, period).start()
    t0 = time.perf_counter()
    try:
        out = fn()
    finally:
        rest = min_seconds - (time.perf_counter() - t0)
        if rest > 0:
            time.sleep(rest)
        summ = s.stop()
    return out, summ
