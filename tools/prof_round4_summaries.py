"""Summaries of tools/prof_round4.sh's counter passes (written under gpurun_out/, copied into profiles/ by the script):
  r4_k5_counters.txt          K5 (quotient_air_kernel<AIR>) alone, one launch per AIR: VALU lane-instructions per
                              constraint evaluation, HBM bytes fetched against the algorithmic bytes read
  r4_sq_loaded_by_kernel.txt  per kernel family, the LOADED 64-txn run: VALU wave-instructions, MFMAs, share
  r4_hash_sq_counters.txt     leaf hashing 2^21 rows x 8 permutations alone on the chip, per kernel form
FETCH_SIZE is in KiB and counts half on gfx950 (x2; MI355X_MICROARCH.md, HBM section; calibrated by the copy kernel in
profiles/r1_pmc_fetch_write_lde_2e14x2432.csv), WRITE_SIZE is in KiB and exact."""
import collections
import csv
import glob
import os
import re

R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
O = os.path.join(R, "gpurun_out")
HEAD = open(os.path.join(R, ".head_for_profiles")).read().strip() if os.path.exists(os.path.join(R, ".head_for_profiles")) else "?"


def rows(dirs):
    for d in dirs:
        for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
            yield from csv.DictReader(open(f))


K5 = os.environ.get("K5_PREFIX", os.environ.get("PROF_ROUND", "r4"))   # r5: tools/prof_round5_k5.sh
RND = os.environ.get("PROF_ROUND", "r4")   # the prefix of every other pass (tools/prof_round.sh)


def k5_summary():
    cases = {}
    log = os.path.join(O, K5 + "_k5_sq.log")
    for line in open(log) if os.path.exists(log) else []:
        m = re.match(r"counters case: air (\d+) (\w+) rows (\d+) cols (\d+) aux (\d+) constraints (\d+) alg_bytes (\d+)", line)
        if m:
            cases[int(m.group(1))] = dict(name=m.group(2), rows=int(m.group(3)), cols=int(m.group(4)), aux=int(m.group(5)),
                                          constraints=int(m.group(6)), alg=int(m.group(7)))
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows((K5 + "_k5_sq", K5 + "_k5_fetch", K5 + "_k5_write")):
        m = re.search(r"quotient_air_kernel<(\d+)u?>", r["Kernel_Name"])
        if m:
            acc[int(m.group(1))][r["Counter_Name"]] += float(r["Counter_Value"])
        elif "quotient_plonk_hash_kernel" in r["Kernel_Name"]:   # AIR 8's Poseidon-gate pass: part of the same quotient
            acc[8][r["Counter_Name"]] += float(r["Counter_Value"])
            acc[108][r["Counter_Name"]] += float(r["Counter_Value"])
    if not acc:
        return
    with open(os.path.join(O, K5 + "_k5_counters.txt"), "w") as out:
        out.write("# HEAD %s.  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES / FETCH_SIZE / WRITE_SIZE (three passes) -- python "
                  "tools/k5_air_probe.py --counters: ONE quotient_air_kernel<AIR> launch per AIR on random LDE matrices (spread form, "
                  "rate 2; rate 8 for AIR 8).  valu_per_constraint = SQ_INSTS_VALU x 64 lanes / (rows x constraints); fetch_over_algorithmic = "
                  "FETCH_SIZE x 2 KiB / (8 x rows x (columns + aux + constant columns)): 1.0 = every LDE element comes from HBM once\n" % HEAD)
        for air, c in sorted(cases.items()):
            v = acc.get(air, {})
            read_alg = c["alg"] - 16.0 * c["rows"]   # the probe's algorithmic bytes less the two quotient columns written
            fetch = 2 * 1024 * v.get("FETCH_SIZE", 0)
            out.write("%-14s rows=%d cols=%d aux=%d constraints=%d  valu_wave_insts=%.4e valu_per_constraint=%.2f  fetched_MB=%.1f "
                      "algorithmic_read_MB=%.1f fetch_over_algorithmic=%.3f written_MB=%.1f\n"
                      % (c["name"], c["rows"], c["cols"], c["aux"], c["constraints"], v.get("SQ_INSTS_VALU", 0),
                         64.0 * v.get("SQ_INSTS_VALU", 0) / (c["rows"] * c["constraints"]), fetch / 1e6, read_alg / 1e6,
                         fetch / read_alg if read_alg else 0, 1024 * v.get("WRITE_SIZE", 0) / 1e6))
        if 108 in acc:
            c, v = cases[8], acc[108]
            out.write("# of plonk's figures, the Poseidon-gate pass (quotient_plonk_hash_kernel, 123 of the constraints) alone: valu_wave_insts=%.4e "
                      "fetched_MB=%.1f (it re-reads the 135 wires and one constant column: 131/240 of the algorithmic read again)\n"
                      % (v.get("SQ_INSTS_VALU", 0), 2 * 1024 * v.get("FETCH_SIZE", 0) / 1e6))
    print(open(os.path.join(O, K5 + "_k5_counters.txt")).read())


def family(k):
    for name in ("leaf_hash_mx_kernel", "leaf_hash_rows", "leaf_hash", "merkle_level_mx_kernel", "merkle_level", "merkle_subtree",
                 "pow_grind", "perm_batch", "fri_layer_leaf", "ntt", "quotient_air_kernel", "quotient_plonk_hash_kernel", "quotient", "fri_", "openings",
                 "aux_suffix", "keccak_ctl", "synth_", "keccak_trace", "query", "combine", "power_vector", "alpha"):
        if name in k:
            return name
    return k.split("(")[0][-40:]


def loaded_summary():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows((RND + "_sq_loaded1", RND + "_sq_loaded2")):
        acc[family(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    if not acc:
        return
    with open(os.path.join(O, RND + "_sq_loaded_by_kernel.txt"), "w") as out:
        out.write("# HEAD %s.  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES (pass 1), "
                  "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES (pass 2) -- python bench.py --txns 64 --steps 1 --warmup 0 "
                  "--no-cpu-baseline --no-profile   (64 txns on 16 prover streams: the loaded chip; counters are summed over the "
                  "launches of a family; under rocprofv3 --pmc kernels are serialised, so these are instruction COUNTS of the "
                  "loaded run's kernel mix, not its timing)\n" % HEAD)
        tot = sum(v.get("SQ_INSTS_VALU", 0) for v in acc.values())
        for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
            out.write("%-28s VALU %.4e (%5.1f %%)  MFMA %.3e  waves %.3e  active_inst_valu %.3e  busy_cycles %.3e\n" % (
                k, v.get("SQ_INSTS_VALU", 0), 100 * v.get("SQ_INSTS_VALU", 0) / max(tot, 1), v.get("SQ_INSTS_MFMA", 0),
                v.get("SQ_WAVES", 0), v.get("SQ_ACTIVE_INST_VALU", 0), v.get("SQ_BUSY_CYCLES", 0)))
        out.write("total VALU wave-instructions %.5e for 64 txn proofs + 63 aggregations + 1 block proof + bp_state_build\n" % tot)
        out.write("total MFMA %.5e\n" % sum(v.get("SQ_INSTS_MFMA", 0) for v in acc.values()))
    print(open(os.path.join(O, RND + "_sq_loaded_by_kernel.txt")).read())


def hash_summary():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows((RND + "_hash_sq1", RND + "_hash_sq2")):
        k = r["Kernel_Name"]
        if "leaf_hash" in k:
            m = re.search(r"leaf_hash_mx_kernel<(\d+), (\d+)>", k) or re.search(r"leaf_hash_mx_kernelILi(\d+)ELi(\d+)E", k)
            name = ("leaf_hash_mx_kernel<4, %s>" % {"0": "per round", "2": "two groups", "3": "three groups"}[m.group(2)]) if m \
                else "leaf_hash_kernel (one lane per state)"
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if not acc:
        return
    with open(os.path.join(O, RND + "_hash_sq_counters.txt"), "w") as out:
        out.write("# HEAD %s.  rocprofv3 --kernel-trace --pmc <SQ counters, two passes> -- python tools/pmc_probe_hash.py: leaf hashing "
                  "of 2^21 rows x 8 permutations = 16777216 permutations per kernel form\n" % HEAD)
        for k, v in acc.items():
            out.write(k + "\n")
            for n, x in sorted(v.items()):
                out.write("    %-28s %.4e\n" % (n, x))
    print(open(os.path.join(O, RND + "_hash_sq_counters.txt")).read())


def leg_trace_summary():
    """r4_kernel_stats_leg_between_markers.csv: per kernel, the launches of bench.py's roofline leg ALONE -- the kernel
    trace of `bench.py --leg-only --leg-skip-extras` restricted to the dispatches between the leg's two marker copies
    (the whole-process stats beside it also hold the state build and the warm-up transaction) -- and, last line, the
    coset-LDE family bench.py's `roofline` is quoted on."""
    files = glob.glob(os.path.join(O, RND + "_leg", "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        return
    trace = list(csv.DictReader(open(files[0])))
    marks = sorted(int(r["Dispatch_Id"]) for r in trace if "calib_copy_u64_kernel" in r["Kernel_Name"])
    if len(marks) != 2:
        print("leg trace: expected two marker dispatches, found", len(marks))
        return
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in trace:
        if marks[0] < int(r["Dispatch_Id"]) < marks[1]:
            a = acc[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    fam = ("ntt16_dit_kernel", "ntt_mx_dit_kernel", "ntt_lds_kernel<false>")
    fn = sum(v[0] for k, v in acc.items() if any(f in k for f in fam))
    ft = sum(v[1] for k, v in acc.items() if any(f in k for f in fam))
    with open(os.path.join(O, RND + "_kernel_stats_leg_between_markers.csv"), "w") as out:
        out.write("# HEAD %s.  rocprofv3 --kernel-trace -- python bench.py --leg-only --leg-skip-extras, dispatches between the leg's marker copies only (2 txn proofs, one prover stream)\n" % HEAD)
        out.write("Name,Calls,TotalDurationNs,AverageNs\n")
        for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            out.write('"%s",%d,%d,%.1f\n' % (k, v[0], v[1], v[1] / v[0]))
        out.write('"coset-LDE family (ntt16_dit_kernel<12|13|14> + ntt_mx_dit_kernel + ntt_lds_kernel<DIT>)",%d,%d,%.1f\n' % (fn, ft, ft / max(fn, 1)))
    print("leg between markers: LDE family %d launches, average %.2f us" % (fn, ft / max(fn, 1) / 1e3))


import sys
k5_summary()
if "k5" not in sys.argv[1:]:
    loaded_summary()
    hash_summary()
    leg_trace_summary()
