"""Shader clock / package power of one GPU sampled on a host thread while device work runs (bench.py).

The fp32-class forward sits on the package power limit (profiles/r02_power_cap.md), so a time without the clock it was
measured at says little: `Sampler` polls the device every `period` seconds — through the in-process `amdsmi` binding
when it imports, else by parsing `rocm-smi --showclocks --showpower` — and `summary()` returns mean / min / max of both
series over the samples taken between start() and stop().

Round 4 (advisor): the sampling thread competes with a launch-bound host thread for the GIL (and the rocm-smi fallback forks),
so bench.py no longer samples INSIDE its timed regions — it repeats the same work in a separate "power pass" right after
each of them and samples that.  The device is identified by its PCI address (torch's device properties -> amdsmi's BDF),
not by the HIP ordinal, which HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES renumber.
"""
import re
import subprocess
import threading
import time


class _AmdSmi:
    name = "amdsmi"

    def __init__(self, index, bdf=None):
        import amdsmi
        self.m = amdsmi
        amdsmi.amdsmi_init()
        hs = amdsmi.amdsmi_get_processor_handles()
        self.h, self.matched = None, "ordinal"
        if bdf is not None:                                 # (domain, bus, device) of the torch device
            for h in hs:
                try:
                    if _parse_bdf(amdsmi.amdsmi_get_gpu_device_bdf(h)) == tuple(bdf):
                        self.h, self.matched = h, "pci"
                        break
                except Exception:
                    pass
        if self.h is None:
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


def _parse_bdf(text):
    """'0000:05:00.0' -> (domain, bus, device)."""
    m = re.match(r"\s*([0-9a-fA-F]+):([0-9a-fA-F]+):([0-9a-fA-F]+)\.", str(text))
    return (int(m.group(1), 16), int(m.group(2), 16), int(m.group(3), 16)) if m else None


def torch_device_bdf(dev):
    """(domain, bus, device) of a torch HIP device, or None when this torch build does not expose it."""
    try:
        import torch
        p = torch.cuda.get_device_properties(dev)
        return (int(p.pci_domain_id), int(p.pci_bus_id), int(p.pci_device_id))
    except Exception:
        return None


class _RocmSmi:
    name = "rocm-smi"
    matched = "ordinal"

    def __init__(self, index, bdf=None):
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
    def __init__(self, device=0, period=0.02):
        """`device`: a torch.device / HIP ordinal.  The smi handle is the one with the same PCI address when both sides expose
        it (`summary()['device_match']` says which rule applied)."""
        self.period = period
        self.src, self.error = None, None
        index = getattr(device, "index", device) or 0
        bdf = torch_device_bdf(device)
        for cls in (_AmdSmi, _RocmSmi):
            try:
                self.src = cls(index, bdf)
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
        return {"source": self.src.name, "device_match": self.src.matched, "samples": len(self.samples), "period_s": self.period,
                "clock_mhz": agg([s[0] for s in self.samples]), "power_w": agg([s[1] for s in self.samples]),
                "power_cap_w": self.src.cap()}


def power_pass(fn, device=0, min_seconds=1.0, period=0.02, sync=None, repetitions=None):
    """Repeat fn() for at least `min_seconds` (or exactly `repetitions` times: ranks of a distributed job must all run the
    same number of collectives) with the sampler running — the way bench.py attaches clock / power figures to a
    timed region WITHOUT sampling inside it: same work, right after it, results discarded.  Returns the summary plus the
    repetitions and the mean seconds per repetition of this pass (`sync()` is called before the clock stops)."""
    s = Sampler(device, period).start()
    t0 = time.perf_counter()
    n = 0
    try:
        while True:
            fn()
            n += 1
            if sync is not None and n % 4 == 0:
                sync()                                      # keep the host at most a few repetitions ahead of the device
            if (n >= repetitions) if repetitions else (time.perf_counter() - t0 >= min_seconds):
                break
        if sync is not None:
            sync()
    finally:
        dt = time.perf_counter() - t0
        summ = s.stop()
    summ["repetitions"] = n
    summ["ms_per_repetition"] = round(dt / max(n, 1) * 1e3, 3)
    summ["sampled"] = "separate pass of the same work right after the timed region (nothing samples inside it)"
    return summ
