"""Per kernel, the SQ counters of tools/prof_round5_ntt.sh's passes (gpurun_out/r5_ntt_pmc*), averaged per launch.
FETCH_SIZE is in KiB and counts half on gfx950 (x2, MI355X_MICROARCH.md HBM section); WRITE_SIZE in KiB, exact."""
import collections
import csv
import glob
import os
import re

R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
O = os.path.join(R, "gpurun_out")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
grid = {}
for f in glob.glob(os.path.join(O, "r5_ntt_pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ntt" not in k:
            continue
        k = re.sub(r"\(bpg.*", "", k.replace("(anonymous namespace)::", "").replace("void ", ""))
        key = "%s grid=%s wg=%s lds=%s vgpr=%s" % (k, r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"), r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?"))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[key][r["Counter_Name"]] += 1
print("# rocprofv3 --kernel-trace --pmc <group> -- python tools/ntt_stall_probe.py, one pass per group; values are per launch (mean)")
print(open(os.path.join(O, "r5_ntt_times.txt")).read() if os.path.exists(os.path.join(O, "r5_ntt_times.txt")) else "")
for key in sorted(acc):
    v = {c: acc[key][c] / cnt[key][c] for c in acc[key]}
    print(key)
    for c in sorted(v):
        print("    %-26s %.4e" % (c, v[c]))
    g = lambda c: v.get(c, 0.0)
    if g("SQ_WAVE_CYCLES"):
        print("    -- per wave-cycle: VALU active %.3f, LDS active %.3f, VMEM active %.3f, wait_inst_any %.3f, wait_inst_lds %.3f, wait_any %.3f"
              % (g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_LDS") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_VMEM") / g("SQ_WAVE_CYCLES"),
                 g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")))
    if g("SQ_BUSY_CYCLES") and g("SQ_WAVES"):
        print("    -- mean waves in flight per SQ-busy cycle x4 SIMDs: %.2f; LDS bank-conflict cycles / LDS active: %.3f; fetched MB %.1f written MB %.1f"
              % (g("SQ_WAVE_CYCLES") / g("SQ_BUSY_CYCLES") if g("SQ_BUSY_CYCLES") else 0, g("SQ_LDS_BANK_CONFLICT") / g("SQ_ACTIVE_INST_LDS") if g("SQ_ACTIVE_INST_LDS") else 0,
                 2 * 1024 * g("FETCH_SIZE") / 1e6, 1024 * g("WRITE_SIZE") / 1e6))
