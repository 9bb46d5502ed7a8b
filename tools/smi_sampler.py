"""Samples the GPU's engine clock, power, temperature and busy percentage from sysfs (hwmon / pp_dpm_sclk /
gpu_busy_percent) at ~20 Hz while a command runs, and prints a per-phase summary: the phases are cut at the
timestamps the command writes to stderr as lines `@phase <name>` (bench.py --phase-marks).

    python tools/smi_sampler.py gpurun_out/smi_A.csv -- python bench.py --phase-marks ...

Reads only world-readable sysfs files of the card the command names (`@pci <domain:bus:dev.fn>` on stderr).
"""
import glob
import os
import subprocess
import sys
import threading
import time


def find_sources(bdf=None):
    """sysfs files of the card whose PCI address is `bdf` (the command names it on stderr as `@pci <bdf>`: a GPU box
    shows all of the host's cards in sysfs, only one of which is ours); the first card when none is named."""
    src = {}
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        if not os.path.exists(os.path.join(dev, "gpu_busy_percent")):
            continue
        if bdf and os.path.basename(os.path.realpath(dev)).lower() != bdf.lower():
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
    src, keys = {}, []
    rows, phases, stop, have_card = [], [], threading.Event(), threading.Event()
    t0 = time.time()

    def sample():
        have_card.wait(120)          # the command names its card first
        while not stop.is_set():
            r = [time.time() - t0]
            for k in keys:
                v = read(src[k])
                r.append(dpm_current(v) if k.startswith("pp_dpm") else v)
            rows.append(r)
            time.sleep(0.05)

    th = threading.Thread(target=sample, daemon=True)
    th.start()
    p = subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)
    for line in p.stderr:
        if line.startswith("@pci ") and not have_card.is_set():
            src.update(find_sources(line.split()[1]) or find_sources())
            keys.extend(sorted(src))
            print("sampling %s" % os.path.dirname(src.get("busy", "?")), file=sys.stderr)
            have_card.set()
        elif line.startswith("@phase "):
            phases.append((time.time() - t0, line.split(None, 1)[1].strip()))
        else:
            sys.stderr.write(line)
    rc = p.wait()
    stop.set()
    have_card.set()
    th.join()
    with open(out_csv, "w") as f:
        f.write(",".join(["t"] + keys) + "\n")
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
