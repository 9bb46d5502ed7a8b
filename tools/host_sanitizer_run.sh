#!/bin/bash
# Host-side AddressSanitizer + UndefinedBehaviorSanitizer pass over the library's CPU code (the CPU verifier, the
# compact-witness decoder, the partial tries, the IR producer, the Poseidon operand tables, argument checking of the
# C ABI) and over the oracle's callers: builds a second libbpg.so with -fsanitize=address,undefined on the HOST code
# only (-fno-gpu-sanitize: GPU sanitizers are not available on this pool), and runs the CPU tests that load it.
# No GPU needed.  Usage: bash tools/host_sanitizer_run.sh [pytest args]   (default: the host-only test files)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${BPG_SAN_DIR:-/tmp/bpg_asan}
mkdir -p "$OUT"
ASANLIB=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
make -C "$R/proof_protocol_decoder_amd/csrc" -j8 OUT="$OUT/libbpg.so" OBJDIR="$OUT/obj" \
  CXXFLAGS="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -Wno-unused-function" > "$OUT/build.log" 2>&1
cd "$R"
TESTS=${@:-tests/test_compact_witness.py tests/test_decoding.py tests/test_trace_protocol.py tests/test_mx_tables.py tests/test_keccak_air.py tests/test_logic_air.py tests/test_memory_air.py tests/test_arithmetic_air.py tests/test_byte_packing_air.py tests/test_keccak_sponge_air.py tests/test_arithmetic_mul_air.py tests/test_plonk_air.py tests/test_lookups.py tests/test_host_cpu.py}
BPG_LIBBPG="$OUT/libbpg.so" LD_PRELOAD="$ASANLIB" ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:verify_asan_link_order=0 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0 python -m pytest $TESTS -q -p no:cacheprovider 2>&1 | tee "$OUT/run.log" | tail -3
N=$(grep -c "runtime error\|AddressSanitizer" "$OUT/run.log" || true)
echo "sanitizer reports: $N (log: $OUT/run.log)"
test "$N" = "0"
