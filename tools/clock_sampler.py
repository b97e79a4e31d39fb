"""Samples the GPU's current sclk / mclk / fclk from sysfs (pp_dpm_*: the starred line) every `period` seconds for
`seconds`, one JSON line per sample on stdout: {"t": unix time, "sclk": MHz, "mclk": MHz, ...}. With --summarise
FILE T0 T1 prints min / median / max of the samples whose time falls inside [T0, T1] (bench.py's `timed_region_unix`).
Development tool for VERDICT r03 item 2(e): is the 6-9 % gap between profiled and unprofiled runs a clock effect?"""
import glob
import json
import re
import statistics
import sys
import time


def read_star(path):
    try:
        for ln in open(path):
            if "*" in ln:
                m = re.search(r"(\d+)\s*Mhz", ln, re.I)
                return int(m.group(1)) if m else None
    except OSError:
        return None
    return None


def sample(seconds, period):
    cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
    t_end = time.time() + seconds
    while time.time() < t_end:
        row = {"t": time.time()}
        for c in cards:
            d = c.rsplit("/", 1)[0]
            tag = d.split("/")[-2]
            for k in ("sclk", "mclk", "fclk", "socclk"):
                v = read_star(d + "/pp_dpm_" + k)
                if v is not None:
                    row[tag + "." + k] = v
        print(json.dumps(row), flush=True)
        time.sleep(period)


def summarise(path, t0, t1):
    rows = [json.loads(ln) for ln in open(path) if ln.strip().startswith("{")]
    inside = [r for r in rows if t0 <= r["t"] <= t1]
    out = {"samples_total": len(rows), "samples_in_region": len(inside)}
    for key in sorted({k for r in rows for k in r if k != "t"}):
        for name, rs in (("region", inside), ("all", rows)):
            vals = [r[key] for r in rs if key in r]
            if vals:
                out["%s[%s]" % (key, name)] = {"min": min(vals), "median": statistics.median(vals), "max": max(vals)}
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2], float(sys.argv[3]), float(sys.argv[4]))
    else:
        sample(float(sys.argv[1]) if len(sys.argv) > 1 else 10.0, float(sys.argv[2]) if len(sys.argv) > 2 else 0.01)
