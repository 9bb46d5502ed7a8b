#!/usr/bin/env python3
"""Extracts the compact-witness golden vectors held by the reference's own tests into
tests/golden/compact_witness_vectors.json (data only: hex payloads and expected state roots).

Sources (read as data, never imported or executed):
  protocol_decoder/src/compact/complex_test_payloads.rs:14-30   TEST_PAYLOAD_1..6 + roots
  protocol_decoder/src/compact/large_test_payloads/test_payload_{5,6}.txt
  protocol_decoder/src/compact/compact_prestate_processing.rs:1439, 1483-1492  SIMPLE_PAYLOAD_STR and
      the six instructions it must parse to
Run in the dev container (needs /root/reference); the GPU box only sees the JSON.
"""
import json
import os
import re

REF = "/root/reference/protocol_decoder/src/compact"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                   "compact_witness_vectors.json")

src = open(os.path.join(REF, "complex_test_payloads.rs")).read()
vectors = []
for n in range(1, 7):
    m = re.search(r"TEST_PAYLOAD_%d: TestProtocolInputAndRoot = TestProtocolInputAndRoot \{(.*?)\};" % n, src, re.S)
    body = m.group(1)
    root = re.search(r'root_str:\s*"([0-9a-f]+)"', body).group(1)
    inc = re.search(r'byte_str:\s*include_str!\("([^"]+)"\)', body)
    if inc:
        payload = open(os.path.join(REF, inc.group(1))).read().strip()
    else:
        payload = re.search(r'byte_str:\s*"([0-9a-f]+)"', body).group(1)
    vectors.append({"name": "complex_payload_%d" % n, "witness_hex": payload, "state_root": root,
                    "source": "complex_test_payloads.rs TEST_PAYLOAD_%d" % n})

proc = open(os.path.join(REF, "compact_prestate_processing.rs")).read()
simple = re.search(r'SIMPLE_PAYLOAD_STR: &str = "([0-9a-f]+)"', proc).group(1)
simple_kat = {
    "witness_hex": simple,
    "source": "compact_prestate_processing.rs:1439,1483-1492",
    # Instruction::Leaf(h_decode_key(k), h_decode(v)) etc., written as plain data
    "instructions": [
        {"op": "leaf", "key_bytes_hex": "10", "value_hex": "31323334"},
        {"op": "leaf", "key_bytes_hex": "10", "value_hex": "31323334"},
        {"op": "branch", "mask": 0b00110000},
        {"op": "leaf", "key_bytes_hex": "0350", "value_hex": "31323335"},
        {"op": "branch", "mask": 0b00011000},
        {"op": "extension", "key_bytes_hex": "0000000000000000000000000000000000000000000000000000000000000012"},
    ],
}
os.makedirs(os.path.dirname(OUT), exist_ok=True)
json.dump({"complex": vectors, "simple": simple_kat}, open(OUT, "w"), indent=1)
print("wrote", OUT, [len(v["witness_hex"]) // 2 for v in vectors], "bytes")
