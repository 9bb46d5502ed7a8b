import numpy as np

P = 0xFFFFFFFF00000001


def rand_field(rng, shape, edge=True):
    """Uniform canonical field elements, with edge values sprinkled in."""
    hi = rng.integers(0, 1 << 32, size=shape, dtype=np.uint64)
    lo = rng.integers(0, 1 << 32, size=shape, dtype=np.uint64)
    x = ((hi << np.uint64(32)) | lo) % np.uint64(P)
    if edge and x.size >= 8:
        flat = x.reshape(-1)
        for k, v in enumerate([0, 1, P - 1, 0xFFFFFFFF, 1 << 32, P - (1 << 32), 0xFFFFFFFF00000000, 2]):
            flat[(k * 7919) % flat.size] = v
    return x


def to_dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def to_host(t):
    return t.cpu().numpy().view(np.uint64)


def bitrev(i, bits):
    r = 0
    for k in range(bits):
        r |= ((i >> k) & 1) << (bits - 1 - k)
    return r


def bitrev_perm(bits):
    i = np.arange(1 << bits, dtype=np.int64)
    r = np.zeros_like(i)
    for k in range(bits):
        r |= ((i >> k) & 1) << (bits - 1 - k)
    return r


def coset_major_to_natural(log_n, rate_bits):
    """index array idx such that natural[i] = coset_major[idx[i]], i = t + 2^r * m <-> t*n + m."""
    n, r = 1 << log_n, 1 << rate_bits
    i = np.arange(n * r)
    return (i % r) * n + (i // r)


def leaf_of_coset_major(log_n, rate_bits):
    """leaf index (upstream reverse_index_bits order) of each coset-major row."""
    n = 1 << log_n
    pos = np.arange(n << rate_bits)
    t, m = pos >> log_n, pos & (n - 1)
    br_n, br_r = bitrev_perm(log_n), bitrev_perm(rate_bits)
    return br_r[t] * n + br_n[m]
