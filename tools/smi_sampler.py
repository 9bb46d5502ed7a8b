"""Samples the GPU's engine clock, power, temperature and busy percentage from sysfs (hwmon / pp_dpm_sclk /
gpu_busy_percent) at ~20 Hz while a command runs, and prints a per-phase summary: the phases are cut at the
timestamps the command writes to stderr as lines `@phase <name>` (bench.py --phase-marks).

    python tools/smi_sampler.py gpurun_out/smi_A.csv -- python bench.py --phase-marks ...

Reads only world-readable sysfs files of card 0's device; falls back to `rocm-smi --json` at 2 Hz when none exist.
"""
import glob
import json
import os
import subprocess
import sys
import threading
import time


def find_sources():
    src = {}
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        if not os.path.exists(os.path.join(dev, "gpu_busy_percent")):
            continue
        src["busy"] = os.path.join(dev, "gpu_busy_percent")
        for name, pat in (("power_uW", "hwmon/hwmon*/power1_average"), ("power_uW", "hwmon/hwmon*/power1_input"),
                          ("temp_mC", "hwmon/hwmon*/temp1_input"), ("temp2_mC", "hwmon/hwmon*/temp2_input"),
                          ("sclk_Hz", "hwmon/hwmon*/freq1_input"), ("mclk_Hz", "hwmon/hwmon*/freq2_input")):
            g = glob.glob(os.path.join(dev, pat))
            if g and name not in src:
                src[name] = g[0]
        for name in ("pp_dpm_sclk", "pp_dpm_mclk"):
            p = os.path.join(dev, name)
            if os.path.exists(p):
                src[name] = p
        break
    return src


def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return ""


def dpm_current(txt):
    for line in txt.splitlines():
        if line.rstrip().endswith("*"):
            return line.split(":")[1].replace("*", "").strip()
    return ""


def main():
    out_csv = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    src = find_sources()
    keys = sorted(src)
    rows, phases, stop = [], [], threading.Event()
    t0 = time.time()

    def sample():
        while not stop.is_set():
            r = [time.time() - t0]
            for k in keys:
                v = read(src[k])
                r.append(dpm_current(v) if k.startswith("pp_dpm") else v)
            rows.append(r)
            time.sleep(0.05)

    def sample_smi():
        while not stop.is_set():
            try:
                j = json.loads(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--showuse", "--json"],
                                              capture_output=True, text=True, timeout=10).stdout)
                rows.append([time.time() - t0, json.dumps(j.get("card0", j))])
            except Exception as e:  # noqa: BLE001
                rows.append([time.time() - t0, "rocm-smi failed: %s" % e])
            time.sleep(0.5)

    th = threading.Thread(target=sample if src else sample_smi, daemon=True)
    th.start()
    p = subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)
    for line in p.stderr:
        if line.startswith("@phase "):
            phases.append((time.time() - t0, line.split(None, 1)[1].strip()))
        else:
            sys.stderr.write(line)
    rc = p.wait()
    stop.set()
    th.join()
    with open(out_csv, "w") as f:
        f.write(",".join(["t"] + (keys if src else ["rocm_smi_json"])) + "\n")
        for r in rows:
            f.write(",".join(str(x).replace(",", ";") for x in r) + "\n")
    if src:
        phases.append((time.time() - t0 + 1, "end"))
        print("phase summary (%s): mean of %s" % (out_csv, keys))
        for (ta, name), (tb, _) in zip(phases, phases[1:]):
            sel = [r for r in rows if ta <= r[0] < tb]
            means = []
            for i, k in enumerate(keys, 1):
                vals = []
                for r in sel:
                    try:
                        vals.append(float(str(r[i]).lower().replace("mhz", "")))
                    except ValueError:
                        pass
                means.append("%s=%.4g" % (k, sum(vals) / len(vals)) if vals else "%s=?" % k)
            print("  %-28s %6.1f s  %s" % (name, tb - ta, "  ".join(means)))
    sys.exit(rc)


if __name__ == "__main__":
    main()
