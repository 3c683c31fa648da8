"""The arithmetic behind csrc/kernels_kpp.hip, checked on the CPU (no library involved): the reference's k-means++ pick is a SEQUENTIAL f32
cumulative sum (src/ivf/index.rs:374-383; the chunk sums of :356-370 likewise), and the device evaluates such a chain in parallel from the
additions' integer images.  While c stays in one binade [2^e, 2^(e+1)) with ulp u, C = c / u is an integer and

    fl(c + x) = u * (C + a + g + (tie and odd(C + a))),   t = x / u,  a = floor(t),  f = t - a,  g = [f > 1/2],  tie = [f == 1/2]

so a RUN of elements is a map 'parity of C -> increment' (d0, d1), runs compose, and only a run whose predicted binade is wrong or that
crosses into the next binade has to be added element by element.  This file restates the kernel's run_summary / composition / chain walk
in numpy scalar arithmetic and compares the result, bit for bit, with the plain sequential sum -- on ties, far-out values, zeros,
denormal-sized values and ragged lengths.  (The GPU side of the same statement: tests/test_gpu_kpp_pick.py.)"""
import struct

import numpy as np
import pytest

f32 = np.float32


def _bits(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def _from_bits(b):
    return f32(struct.unpack("<f", struct.pack("<I", b))[0])


def _summary(xs, e):
    """kernels_kpp.hip: run_summary -- (d0, d1) of the elements under the ulp of biased exponent e."""
    scale = _from_bits((277 - e) << 23)
    limit = _from_bits((e + 1) << 23)
    d = [0, 0]
    for x in xs:
        if x >= limit:
            a, g, tie = 1 << 24, 0, 0
        else:
            t = f32(x * scale)
            fl = np.floor(t)
            fr = f32(t - fl)
            a, g, tie = int(fl), int(fr > f32(0.5)), int(fr == f32(0.5))
        d[0] += a + g + (tie & ((a ^ d[0]) & 1))
        d[1] += a + g + (tie & ((a ^ ~d[1]) & 1))
    return min(d[0], 1 << 25), min(d[1], 1 << 25)


def _then(p, q):
    """kernels_kpp.hip: pair_then -- the pair of two consecutive stretches."""
    return p[0] + (q[1] if p[0] & 1 else q[0]), p[1] + (q[0] if p[1] & 1 else q[1])


def _chain(x, run):
    """The chain over x from 0 with runs of `run` elements: predicted binades from an approximate prefix, the exact check on arrival."""
    n = len(x)
    n_runs = (n + run - 1) // run
    xp = np.zeros(n_runs * run, dtype=np.float32)
    xp[:n] = x
    runs = xp.reshape(n_runs, run)
    approx = np.concatenate([[0.0], np.cumsum(runs.sum(axis=1, dtype=np.float32), dtype=np.float32)[:-1]]).astype(np.float32)
    c = f32(0)
    starts, walked = [], 0
    for r in range(n_runs):
        starts.append(c)
        e_pred = (_bits(approx[r]) >> 23) & 0x1FF
        cb = _bits(c)
        ec = cb >> 23
        if 23 <= e_pred < 254 and e_pred == ec:
            C = (cb & 0x7FFFFF) | 0x800000
            Cn = C + _summary(runs[r], e_pred)[C & 1]
            if Cn < (1 << 24):
                c = _from_bits((ec << 23) | (Cn & 0x7FFFFF))
                continue
        walked += 1
        for v in runs[r]:
            c = f32(c + v)
    return c, starts, walked


def _cases(rng):
    yield "uniform", rng.random(3000, dtype=np.float32) * f32(100)
    yield "integer-valued (a tie in almost every add once c >= 2^24)", rng.integers(0, 1 << 18, 2500).astype(np.float32)
    yield "quarters", (rng.integers(0, 1 << 12, 2000) * 0.25).astype(np.float32)
    yield "mostly zero", np.where(rng.random(2000) < 0.9, 0, rng.random(2000) * 1e6).astype(np.float32)
    x = (rng.random(2000) ** 8 * 1e-3).astype(np.float32)
    x[rng.integers(0, 2000, 3)] = f32(1e10)
    yield "far-out values", x
    yield "denormal-sized", (rng.random(1500) * 1e-38).astype(np.float32)
    yield "huge", (rng.random(1500) * 1e33).astype(np.float32)
    yield "one value", np.full(1777, 3.0, dtype=np.float32)
    yield "powers of two", np.exp2(rng.integers(-20, 20, 1500)).astype(np.float32)


@pytest.mark.parametrize("run", [8, 14, 56])
def test_composed_integer_images_reproduce_the_sequential_f32_sum(run):
    rng = np.random.default_rng(56 + run)
    for name, x in _cases(rng):
        ref = np.cumsum(x, dtype=np.float32)             # numpy's accumulate is the sequential chain (no re-association)
        c, starts, walked = _chain(x, run)
        assert _bits(c) == _bits(ref[-1]), (name, run)
        for r in range(1, len(starts)):
            assert _bits(starts[r]) == _bits(ref[r * run - 1]), (name, run, r)
        # the element-wise fallback is the exception: about one run per binade the sum passes through, plus mispredictions
        if name == "uniform":
            assert walked <= 24, (walked, len(starts))


def test_pairs_compose_associatively():
    rng = np.random.default_rng(7)
    for _ in range(300):
        p, q, r = [(int(a), int(a) + int(t)) for a, t in zip(rng.integers(0, 1 << 20, 3), rng.integers(-1, 2, 3))]
        p, q, r = [(max(a, 0), max(b, 0)) for a, b in (p, q, r)]
        assert _then(_then(p, q), r) == _then(p, _then(q, r))
