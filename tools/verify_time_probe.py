import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from proof_protocol_decoder_amd import proof_gen as pg
from proof_protocol_decoder_amd.block_driver import synthetic_block_irs
S1_LOG_N = (16, 9, 12, 14, 9, 12, 17); S1_WIDTH = (128, 128, 192, 2432, 512, 320, 16)
st = pg.ProverStateBuilder().set(n_workers=2, arena_bytes=5 << 30).build()
irs = synthetic_block_irs(5, 2, S1_LOG_N, S1_WIDTH)
t = [pg.generate_txn_proof(st, ir) for ir in irs]
t0 = time.time(); a = pg.generate_agg_proof(st, t[0], t[1]); t_agg = time.time() - t0
t0 = time.time(); a = pg.generate_agg_proof(st, t[0], t[1]); t_agg2 = time.time() - t0
v = pg.VerifierState.from_prover_state(st)
t0 = time.time()
for _ in range(50): v.verify_any(a.intern)
tv = (time.time() - t0) / 50
t0 = time.time(); b = pg.generate_block_proof(st, None, a); t_blk = time.time() - t0
print("agg proof (lone, incl. 2 child verifications) %.1f ms / %.1f ms; CPU verification of one recursion-shaped proof %.2f ms; block proof %.1f ms" % (t_agg * 1e3, t_agg2 * 1e3, tv * 1e3, t_blk * 1e3))
st.close()
