"""`__graft_entry__.build()` is the driver's does-it-build check: every HIP source (library, tools) must
cross-compile for gfx950 and the package must load.  CPU only (hipcc needs no GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_entry_compiles_everything():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()
    for rel in ("proof_protocol_decoder_amd/lib/libbpg.so", "oracle/liboracle.so", "tools/microbench", "tools/wait_probe"):
        assert os.path.exists(os.path.join(ROOT, rel)), rel
