"""Shader clock / package power of one GPU sampled on a host thread while device work runs (bench.py).

The fp32-class forward sits on the package power limit (profiles/r02_power_cap.md), so a time without the clock it was
measured at says little: `Sampler` polls the device every `period` seconds — through the in-process `amdsmi` binding
when it imports, else by parsing `rocm-smi --showclocks --showpower` — and `summary()` returns mean / min / max of both
series over the samples taken between start() and stop().

Round 5: one sample is the device's whole gpu_metrics table — the shader clock of every XCD, the memory / SoC / fabric
clocks, temperatures, and the firmware's own limiter residency accumulators (PPT power, PROCHOT, socket / VR / HBM thermal;
per-XCD "clock below the host limit because of power / temperature") — and the socket energy counter is read at start() and
stop(): `summary()['limiter']` names the throttle reason the DEVICE reports instead of inferring it from watts, and
`power_from_energy_w` is a mean power that does not depend on 20-ms point samples.

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
    # gpu_metrics residency accumulators (PMFW samples in which a limiter was active) -> key in summary()['limiter']
    RESIDENCY = (("ppt_residency_acc", "ppt_power"), ("prochot_residency_acc", "prochot_thermal"),
                 ("socket_thm_residency_acc", "socket_thermal"), ("vr_thm_residency_acc", "vr_thermal"),
                 ("hbm_thm_residency_acc", "hbm_thermal"))

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
        self.metrics_ok = True
        self.read()                                         # fail here, not on the sampling thread

    @staticmethod
    def _num(d, keys):
        for k in keys:
            v = d.get(k) if isinstance(d, dict) else None
            if isinstance(v, (int, float)) and not isinstance(v, bool) and v == v and v > 0:
                return float(v)
        return None

    @staticmethod
    def _count(d, k):
        """An accumulator: 0 is a value (unlike a clock or a wattage), 'N/A' / max-uint placeholders are not."""
        v = d.get(k) if isinstance(d, dict) else None
        return int(v) if isinstance(v, int) and not isinstance(v, bool) and 0 <= v < (1 << 63) else None

    def _metrics(self):
        """One amdsmi_get_gpu_metrics_info() call = the whole PMFW table: per-XCD shader clocks, memory (uclk) / SoC clocks,
        socket power, temperatures, the limiter residency accumulators and the per-XCD "clock below the host limit because of
        power / temperature" accumulators.  None when this binding / firmware has no such table."""
        if not self.metrics_ok:
            return None
        try:
            g = self.m.amdsmi_get_gpu_metrics_info(self.h)
        except Exception:
            self.metrics_ok = False
            return None
        out = {}
        xcd = [float(v) for v in (g.get("current_gfxclks") or []) if isinstance(v, (int, float)) and 0 < v < 10000]
        if xcd:
            out["xcd_clocks"] = xcd
        for src, dst in (("current_uclk", "mem_clock"), ("current_socclk", "soc_clock"), ("current_socket_power", "metrics_power"),
                         ("temperature_hotspot", "temp_hotspot"), ("temperature_mem", "temp_mem"),
                         ("average_gfxclk_frequency", "avg_gfx_clock")):
            v = self._num(g, (src,))
            if v is not None and v < 60000:
                out[dst] = v
        if "soc_clock" not in out:
            soc = [float(v) for v in (g.get("current_socclks") or []) if isinstance(v, (int, float)) and 0 < v < 10000]
            if soc:
                out["soc_clock"] = sum(soc) / len(soc)
        acc = {k: self._count(g, k) for k in ("accumulation_counter",) + tuple(k for k, _ in self.RESIDENCY)}
        if acc["accumulation_counter"] is not None:
            out["acc"] = acc
        for src, dst in (("xcp_stats.gfx_below_host_limit_ppt_acc", "below_host_limit_ppt"),
                         ("xcp_stats.gfx_below_host_limit_thm_acc", "below_host_limit_thm"),
                         ("xcp_stats.gfx_below_host_limit_total_acc", "below_host_limit_total"),
                         ("xcp_stats.gfx_low_utilization_acc", "low_utilization")):
            rows = g.get(src)
            try:                                            # [partition][xcd] counters; partition 0 holds the device in SPX mode
                vals = [int(v) for v in rows[0] if isinstance(v, int) and 0 <= v < (1 << 63)]
                if vals:
                    out[dst] = vals
            except Exception:
                pass
        for k in ("throttle_status", "indep_throttle_status"):
            v = g.get(k)
            if isinstance(v, (int, bool)):
                out[k] = int(v)
        return out

    def _clk(self, kind):
        try:
            return self._num(self.m.amdsmi_get_clock_info(self.h, kind), ("clk", "cur_clk", "current_clk"))
        except Exception:
            return None

    def read(self):
        """One sample: {'clock', 'power'} always (shader clock MHz, socket W; None when a field is missing) plus whatever the
        gpu_metrics table and the other clock domains offer."""
        m = self.m
        p = m.amdsmi_get_power_info(self.h)
        power = self._num(p, ("current_socket_power", "average_socket_power", "socket_power"))
        clock = self._clk(m.AmdSmiClkType.GFX)
        if power is None and clock is None:
            raise RuntimeError(f"amdsmi returned no usable fields: {p}")
        s = {"clock": clock, "power": power}
        g = self._metrics()
        if g:
            s.update(g)
        if "mem_clock" not in s:
            s["mem_clock"] = self._clk(m.AmdSmiClkType.MEM)
        s["fabric_clock"] = self._clk(m.AmdSmiClkType.DF)
        return s

    def energy(self):
        """(accumulator counts, joules per count) of the socket energy counter, or None."""
        try:
            e = self.m.amdsmi_get_energy_count(self.h)
            res = float(e["counter_resolution"])            # micro-joules per count
            return int(e["energy_accumulator"]), res * 1e-6
        except Exception:
            return None

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
        return {"clock": float(c.group(1)) if c else None, "power": float(p.group(1)) if p else None}

    def energy(self):
        return None

    def cap(self):
        return None


def device_index(device):
    """HIP ordinal of a torch.device / "cuda:1" / "cuda" / int / None (advisor r04: a str used to resolve to the bound method
    `str.index`, the TypeError was swallowed and the sampler silently reported nothing)."""
    if device is None:
        return 0
    if isinstance(device, int):
        return device
    if isinstance(device, str):
        m = re.match(r"^\s*(?:cuda|hip)?:?(\d+)?\s*$", device)
        if not m:
            raise ValueError(f"not a GPU device: {device!r}")
        return int(m.group(1) or 0)
    idx = getattr(device, "index", None)
    return int(idx) if isinstance(idx, int) else 0


def _agg(vals, digits=1):
    vals = [v for v in vals if v is not None]
    if not vals:
        return None
    return {"mean": round(sum(vals) / len(vals), digits), "min": round(min(vals), digits), "max": round(max(vals), digits)}


class Sampler:
    def __init__(self, device=0, period=0.02):
        """`device`: a torch.device / "cuda:N" / HIP ordinal.  The smi handle is the one with the same PCI address when both
        sides expose it (`summary()['device_match']` says which rule applied)."""
        self.period = period
        self.src, self.error = None, None
        index = device_index(device)
        bdf = torch_device_bdf(index)
        for cls in (_AmdSmi, _RocmSmi):
            try:
                self.src = cls(index, bdf)
                break
            except Exception as e:                          # no GPU / no permission / unknown fields: report, never raise
                self.error = f"{cls.name}: {type(e).__name__}: {e}"[:200]
        self._stop = threading.Event()
        self._thread = None
        self.samples = []
        self._e0 = self._e1 = None
        self._t0 = self._t1 = None

    def start(self):
        self.samples = []
        if self.src is None:
            return self
        self._stop.clear()
        self._e0, self._t0 = self.src.energy(), time.perf_counter()
        self._e1 = None

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
            self._e1, self._t1 = self.src.energy(), time.perf_counter()
        return self.summary()

    def _limiter(self):
        """Which limiter the DEVICE says was active between the first and the last sample: the share of the firmware's own
        samples (gpu_metrics `accumulation_counter`) in which each residency accumulator advanced, plus the per-XCD "shader clock
        below the host limit because of power / temperature" counters.  'unavailable' when the table lacks them."""
        acc = [s["acc"] for s in self.samples if isinstance(s.get("acc"), dict)]
        if len(acc) < 2:
            return {"source": "unavailable", "note": "this amdsmi binding / firmware exposes no residency accumulators"}
        a, b = acc[0], acc[-1]
        n = b["accumulation_counter"] - a["accumulation_counter"]
        out = {"source": "amdsmi gpu_metrics residency accumulators, last sample minus first", "firmware_samples": n}
        if n <= 0:
            out["source"] = "unavailable"
            out["note"] = "accumulation_counter did not advance between the first and the last sample"
            return out
        fr = {}
        for key, name in _AmdSmi.RESIDENCY:
            if a.get(key) is not None and b.get(key) is not None:
                fr[name] = round((b[key] - a[key]) / n, 4)
        out["active_frac"] = fr
        for key in ("below_host_limit_ppt", "below_host_limit_thm", "below_host_limit_total", "low_utilization"):
            rows = [s[key] for s in self.samples if isinstance(s.get(key), list)]
            if len(rows) >= 2 and len(rows[0]) == len(rows[-1]):
                d = [(y - x) / n for x, y in zip(rows[0], rows[-1])]
                out["xcd_" + key + "_frac"] = {"mean": round(sum(d) / len(d), 4), "min": round(min(d), 4), "max": round(max(d), 4)}
        top = max(fr.items(), key=lambda kv: kv[1]) if fr else (None, 0.0)
        out["dominant"] = top[0] if top[1] >= 0.05 else None
        bits = [s[k] for s in self.samples for k in ("throttle_status", "indep_throttle_status") if s.get(k) is not None]
        if bits:
            out["throttle_status_or"] = int(max(bits))
        return out

    def summary(self):
        if self.src is None:
            return {"source": None, "samples": 0, "error": self.error}
        S = self.samples
        out = {"source": self.src.name, "device_match": self.src.matched, "samples": len(S), "period_s": self.period,
               "clock_mhz": _agg([s.get("clock") for s in S]), "power_w": _agg([s.get("power") for s in S]),
               "power_cap_w": self.src.cap()}
        # the other clock domains: memory (uclk), SoC, data fabric; and the shader clock of EVERY XCD (the GFX clock above is
        # one number): mean over samples of the per-sample mean / min / max across XCDs
        for key, name in (("mem_clock", "mem_clock_mhz"), ("soc_clock", "soc_clock_mhz"), ("fabric_clock", "fabric_clock_mhz"),
                          ("temp_hotspot", "temp_hotspot_c"), ("temp_mem", "temp_mem_c")):
            a = _agg([s.get(key) for s in S])
            if a is not None:
                out[name] = a
        xs = [s["xcd_clocks"] for s in S if s.get("xcd_clocks")]
        if xs:
            out["xcd_clock_mhz"] = {"xcds": len(xs[0]), "mean": round(sum(sum(x) / len(x) for x in xs) / len(xs), 1),
                                    "slowest_xcd_mean": round(sum(min(x) for x in xs) / len(xs), 1),
                                    "fastest_xcd_mean": round(sum(max(x) for x in xs) / len(xs), 1),
                                    "per_xcd_mean": [round(sum(x[i] for x in xs) / len(xs), 1) for i in range(len(xs[0]))]}
            # which XCD is amdsmi's single GFX `clk`?  (two calls a moment apart: nearest XCD per sample, counted)
            near = {}
            for s_ in S:
                if s_.get("clock") and s_.get("xcd_clocks"):
                    i = min(range(len(s_["xcd_clocks"])), key=lambda k: abs(s_["xcd_clocks"][k] - s_["clock"]))
                    near[i] = near.get(i, 0) + 1
            if near:
                out["xcd_clock_mhz"]["gfx_clk_nearest_xcd_counts"] = {str(k): v for k, v in sorted(near.items())}
        out["limiter"] = self._limiter()
        if self._e0 and self._e1 and self._t1 and self._t1 > self._t0:
            joules = (self._e1[0] - self._e0[0]) * self._e1[1]
            if joules > 0:
                out["energy_j"] = round(joules, 2)
                out["power_from_energy_w"] = round(joules / (self._t1 - self._t0), 1)
        return out


def power_pass(fn, device=0, min_seconds=1.0, period=0.02, sync=None, repetitions=None):
    """Repeat fn() for at least `min_seconds` (or exactly `repetitions` times: ranks of a distributed job must all run the
    same number of collectives) with the sampler running — the way bench.py attaches clock / power figures to a
    timed region WITHOUT sampling inside it: same work, right after it, results discarded.  Returns the summary plus the
    repetitions and the mean seconds per repetition of this pass (`sync()` is called before the clock stops).
    NOTE: `fn` really runs — a `fn` that trains keeps stepping the optimizer (bench.py's training leg relies on that and
    zeroes the weights afterwards); pass an idempotent `fn` when the model's state matters to the caller."""
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
    if summ.get("energy_j"):
        summ["energy_j_per_repetition"] = round(summ["energy_j"] / max(n, 1), 3)
    summ["sampled"] = "separate pass of the same work right after the timed region (nothing samples inside it)"
    return summ
