"""HBM traffic per launch of the coset-LDE kernel family in bench.py's single-stream `roofline` leg,
from two rocprofv3 counter passes (MI355X_MICROARCH.md, HBM section: FETCH_SIZE and WRITE_SIZE in
separate passes, both in KiB; FETCH_SIZE counts half on gfx950 -> x2, see the calibration copy in
profiles/r1_pmc_fetch_write_lde_2e14x2432.csv; WRITE_SIZE is exact).

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- \\
      python bench.py --txns 2 --threads 1 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_f.log
  (same with WRITE_SIZE into gpurun_out/pmc_w)
  python tools/pmc_family_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_f.log

bench.py brackets its single-stream roofline leg with two one-word `calib_copy_u64_kernel` marker
dispatches; the family launches between them are the leg, and their algorithmic bytes come from the
JSON line of the same run."""
import csv, glob, json, os, sys

FAMILY = ("ntt16_dit_kernel", "ntt_mx_dit_kernel", "ntt_lds_kernel<false>")


def family_rows(d, counter):
    rows, marks = [], []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True) or [d]:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            if "calib_copy_u64_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) > 0:
                marks.append(int(r["Dispatch_Id"]))
            elif any(k in r["Kernel_Name"] for k in FAMILY):
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    assert len(marks) == 2, "expected the two marker dispatches of bench.py's roofline leg, got %d" % len(marks)
    lo, hi = sorted(marks)
    return sorted(x for x in rows if lo < x[0] < hi)


def main():
    line = [l for l in open(sys.argv[3]) if l.startswith("{")][-1]
    roof = json.loads(line)["roofline"]
    n, alg = roof["launches"], roof["alg_bytes_per_launch"]
    fetch = family_rows(sys.argv[1], "FETCH_SIZE")
    write = family_rows(sys.argv[2], "WRITE_SIZE")
    assert len(fetch) == n and len(write) == n, (len(fetch), len(write), n)
    fb = 2 * 1024 * sum(v for _, v in fetch) / n
    wb = 1024 * sum(v for _, v in write) / n
    print("roofline leg: %d launches, algorithmic %.2f MB/launch" % (n, alg / 1e6))
    print("fetched %.2f MB/launch (FETCH_SIZE x2), written %.2f MB/launch (WRITE_SIZE)" % (fb / 1e6, wb / 1e6))
    print("HBM traffic %.2f MB/launch = %.3f x algorithmic" % ((fb + wb) / 1e6, (fb + wb) / alg))


if __name__ == "__main__":
    main()
