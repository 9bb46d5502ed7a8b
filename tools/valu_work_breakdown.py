"""VALU work per kernel of one synthetic txn proof, from `rocprofv3 --pmc SQ_INSTS_VALU` (wave-instructions):
where the chip's issue slots go when the block run is VALU-bound.
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d OUT -- python bench.py --txns 4 --threads 1 \
      --steps 1 --warmup 0 --no-cpu-baseline --no-profile --quad-threshold-log2 13
  python tools/valu_work_breakdown.py OUT 4
(--quad-threshold-log2 13 = the kernel-form choice of the loaded chip)."""
import collections, csv, glob, os, re, sys

d, n_txn = sys.argv[1], int(sys.argv[2])
f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
tot, calls = collections.Counter(), collections.Counter()
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "SQ_INSTS_VALU":
        continue
    k = re.sub(r"\(anonymous namespace\)::|bpg::|void |\(.*$", "", r["Kernel_Name"])
    tot[k] += float(r["Counter_Value"])
    calls[k] += 1
s = sum(tot.values())
print("VALU wave-instructions per txn: %.3e  (x 4.4 cycles / (1024 SIMDs x 2.4 GHz) = %.1f ms of the whole chip)"
      % (s / n_txn, s / n_txn * 4.4 / (1024 * 2.4e9) * 1e3))
for k, v in tot.most_common(25):
    print("%-44s %8d launches  %10.3e  %5.1f %%" % (k[:44], calls[k] / n_txn, v / n_txn, 100 * v / s))
