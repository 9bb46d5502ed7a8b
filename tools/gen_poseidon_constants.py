#!/usr/bin/env python3
"""Regenerate the 360 Poseidon-Goldilocks round constants and check them against KATs.

Provenance (read this before trusting the table):
  The arithmetic under the hot path lives in plonky2 @ 265d46a9, which is NOT in
  /root/reference, so the constants cannot be copied or diffed against the reference.  They are
  re-derived here, and the derivation was FOUND BY SEARCH, not known in advance.  Three candidate
  procedures were tried, in this order:
    1. Grain LFSR of eprint 2019/458 (field=1, sbox=0, n=64, t=12, R_F=8, R_P=22) -- the
       hypothesis recorded in SURVEY.md Appendix A.  REJECTED: zero-state KAT mismatch.
    2. ChaCha8Rng::seed_from_u64(0) + rand-0.8 gen_range(0..p) with the Crandall prime
       p = 2^64 - 9*2^28 + 1.  Reproduces an older constant set (first four recalled constants
       0xb585f767417ee042, ...), REJECTED for this field: fifth/sixth recalled constants differ.
    3. ChaCha8Rng::seed_from_u64(0) + rand-0.8 gen_range(0..p), p = 2^64 - 2^32 + 1.  ACCEPTED.
  Evidence for (3): with these constants, the MDS below, x^7 and 4+22+4 rounds, the permutation
  reproduces BOTH recalled upstream test vectors on all 12 output words -- the all-zero input
  (its first four words are the KAT in SURVEY.md Appendix A; this vector was used to select the
  procedure) and the input (0,1,...,11) (not used for selection).  The KATs themselves are
  recalled from public knowledge of plonky2, i.e. [UPSTREAM-UNVERIFIED]; nothing in
  /root/reference pins them.

Usage: gen_poseidon_constants.py [out.inc ...]   (always runs the KAT self-check first)
"""
import sys

P = 0xFFFFFFFF00000001
T = 12
R_F = 8
R_P = 22
N_ROUNDS = R_F + R_P
MDS_CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
MDS_DIAG = [8] + [0] * 11
M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF

KAT = [
    ([0] * 12,
     [0x3c18a9786cb0b359, 0xc4055e3364a246c3, 0x7953db0ab48808f4, 0xc71603f33a1144ca,
      0xd7709673896996dc, 0x46a84e87642f44ed, 0xd032648251ee0b3c, 0x1c687363b207df62,
      0xdf8565563e8045fe, 0x40f5b37ff4254dae, 0xd070f637b431067c, 0x1792b1c4342109d7]),
    (list(range(12)),
     [0xd64e1e3efc5b8e9e, 0x53666633020aaa47, 0xd40285597c6a8825, 0x613a4f81e81231d2,
      0x414754bfebd051f0, 0xcb1f8980294a023f, 0x6eb2a9e4d54a9d0f, 0x1902bc3af467e056,
      0xf045d5eafdc6021f, 0xe4150f77caaa3be5, 0xc9bfd01d39b50cce, 0x5c0a27fcb0e1459b]),
]


def _rotl(x, n):
    return ((x << n) & M32) | (x >> (32 - n))


def _qr(s, a, b, c, d):
    s[a] = (s[a] + s[b]) & M32; s[d] = _rotl(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & M32; s[b] = _rotl(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b]) & M32; s[d] = _rotl(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & M32; s[b] = _rotl(s[b] ^ s[c], 7)


def _chacha8_block(key, counter):
    st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + key + [counter & M32, counter >> 32, 0, 0]
    w = st[:]
    for _ in range(4):  # 8 rounds
        _qr(w, 0, 4, 8, 12); _qr(w, 1, 5, 9, 13); _qr(w, 2, 6, 10, 14); _qr(w, 3, 7, 11, 15)
        _qr(w, 0, 5, 10, 15); _qr(w, 1, 6, 11, 12); _qr(w, 2, 7, 8, 13); _qr(w, 3, 4, 9, 14)
    return [(w[i] + st[i]) & M32 for i in range(16)]


def _seed_from_u64(state):
    """rand_core SeedableRng::seed_from_u64: PCG32 expands the u64 into the 32-byte key."""
    key = []
    for _ in range(8):
        state = (state * 6364136223846793005 + 11634580027462260723) & M64
        xs = (((state >> 18) ^ state) >> 27) & M32
        rot = state >> 59
        key.append(((xs >> rot) | (xs << ((32 - rot) & 31))) & M32)
    return key


class ChaCha8Rng:
    def __init__(self, seed):
        self.key = _seed_from_u64(seed)
        self.ctr = 0
        self.buf = []

    def next_u64(self):
        if len(self.buf) < 2:
            self.buf += _chacha8_block(self.key, self.ctr)
            self.ctr += 1
        lo = self.buf.pop(0)
        hi = self.buf.pop(0)
        return (hi << 32) | lo

    def gen_range(self, n):
        """rand 0.8 UniformInt::sample_single(0, n): widening multiply with rejection zone."""
        zone = ((n << (64 - n.bit_length())) - 1) & M64
        while True:
            prod = self.next_u64() * n
            if (prod & M64) <= zone:
                return prod >> 64


def round_constants():
    rng = ChaCha8Rng(0)
    return [rng.gen_range(P) for _ in range(N_ROUNDS * T)]


def mds(state):
    out = []
    for r in range(T):
        acc = sum(state[(i + r) % T] * MDS_CIRC[i] for i in range(T)) + state[r] * MDS_DIAG[r]
        out.append(acc % P)
    return out


def permute(state, rc):
    state = [x % P for x in state]
    rnd = 0
    for full, n_rounds in ((True, R_F // 2), (False, R_P), (True, R_F // 2)):
        for _ in range(n_rounds):
            state = [(state[i] + rc[rnd * T + i]) % P for i in range(T)]
            if full:
                state = [pow(x, 7, P) for x in state]
            else:
                state[0] = pow(state[0], 7, P)
            state = mds(state)
            rnd += 1
    return state


def self_check(rc):
    for inp, want in KAT:
        got = permute(inp, rc)
        if got != want:
            raise SystemExit("Poseidon KAT mismatch for input %r" % (inp,))


if __name__ == "__main__":
    rc = round_constants()
    self_check(rc)
    print("KATs ok (zero state and 0..11, 12 words each); rc[0]=%#x max=%#x" % (rc[0], max(rc)),
          file=sys.stderr)
    for path in sys.argv[1:]:
        with open(path, "w") as f:
            f.write("/* Generated by tools/gen_poseidon_constants.py -- ChaCha8Rng::seed_from_u64(0),\n"
                    " * gen_range(0..p); see that file's docstring for provenance. 30 rounds x 12. */\n")
            for i in range(0, len(rc), 4):
                f.write("  " + ", ".join("0x%016xULL" % x for x in rc[i:i + 4]) + ",\n")
