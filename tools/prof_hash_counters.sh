set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/r2b_hash_sq1 -- python $R/tools/pmc_probe_hash.py > $O/r2b_hash_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/r2b_hash_sq2 -- python $R/tools/pmc_probe_hash.py > $O/r2b_hash_sq2.log 2>&1
cd $R && python - <<'PY'
import csv, glob, os, collections
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for d in ("r2b_hash_sq1", "r2b_hash_sq2"):
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "leaf_hash" in k:
                name = "leaf_hash_mx_kernel<4>" if "mx" in k else "leaf_hash_kernel (one lane per state)"
                acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
with open(os.path.join(O, "r2b_hash_sq_counters.txt"), "w") as out:
    out.write("# rocprofv3 --kernel-trace --pmc <SQ counters, two passes> -- python tools/pmc_probe_hash.py: leaf hashing of 2^21 rows x 8 permutations\n")
    for k, v in acc.items():
        out.write(k + "\n")
        for n, x in sorted(v.items()):
            out.write("    %-28s %.4e\n" % (n, x))
print(open(os.path.join(O, "r2b_hash_sq_counters.txt")).read())
PY
