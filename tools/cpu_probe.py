"""Wall time vs CPU time of a bench.py run (are the prover threads sleeping or spinning while they wait?).
usage: python tools/cpu_probe.py [bench.py args...]"""
import resource, subprocess, sys, time
args = sys.argv[1:] or ["--txns", "128", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-profile"]
t0 = time.time()
p = subprocess.run([sys.executable, "bench.py"] + args, capture_output=True, text=True)
dt = time.time() - t0
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
import json
try:
    v = json.loads(p.stdout.strip().splitlines()[-1])["value"]
except Exception:
    v = p.stdout[-200:] + p.stderr[-400:]
print("%s -> %s txn-proofs/s; wall %.1f s, user %.1f s, sys %.1f s, (user+sys)/wall = %.1f cores; ctx switches vol %d invol %d" % (
    " ".join(args), v, dt, ru.ru_utime, ru.ru_stime, (ru.ru_utime + ru.ru_stime) / dt, ru.ru_nvcsw, ru.ru_nivcsw), flush=True)
