"""The k-means++ pick on the device (kernels_kpp.hip) against the reference's definition (src/ivf/index.rs:354-390):
    total     = the worker chunks' SEQUENTIAL f32 sums joined in ascending chunk order (:356-370, chunking :259-265)
    threshold = draw * total (:373)
    pick      = the first slot whose SEQUENTIAL f32 cumulative sum is >= threshold (:374-383)
numpy's add.accumulate in float32 is that sequential chain (np.cumsum never re-associates).  The kernel evaluates the chains in parallel
by composing the additions' integer images; the bar is bit equality of the total and equality of the pick on every input -- ties of
round-to-nearest-even (integer-valued data), values far above the running sum, zeros, denormals, ragged lengths."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reference(md, workers, draw):
    n = len(md)
    w = max(1, min(workers, n))
    chunk = (n + w - 1) // w
    total = np.float32(0)
    for s in range(0, n, chunk):
        total = np.float32(total + np.cumsum(md[s:s + chunk], dtype=np.float32)[-1])
    thr = np.float32(np.float32(draw) * total)
    cs = np.cumsum(md, dtype=np.float32)
    hit = np.nonzero(cs >= thr)[0]
    return total, (int(hit[0]) if len(hit) else None)


def _device(pqv, md, workers, draw):
    from pq_vector_amd import _ffi
    md = np.ascontiguousarray(md, dtype=np.float32)
    pick = C.c_uint64(2 ** 64 - 1); total = C.c_float(0); status = C.c_uint32(99)
    rc = _ffi.lib().pqv_kpp_pick(0, md.ctypes.data_as(C.POINTER(C.c_float)), len(md), workers, C.c_float(draw), C.byref(pick),
                                 C.byref(total), C.byref(status))
    assert rc == 0
    return np.float32(total.value), int(pick.value), int(status.value)


def _cases(rng):
    for n in (1, 2, 55, 56, 57, 63, 64, 65, 1000, 3584, 3585, 6250, 8191, 8193, 50_000, 57_343, 57_344):
        yield "uniform", rng.random(n, dtype=np.float32) * np.float32(100)
    for n in (777, 20_000, 50_000):
        yield "squared distances of integer data", rng.integers(0, 1 << 18, n).astype(np.float32)          # ties in every run once c >= 2^24
        yield "quarters", (rng.integers(0, 1 << 12, n) * 0.25).astype(np.float32)
        yield "mostly zero", np.where(rng.random(n) < 0.9, 0, rng.random(n) * 1e6).astype(np.float32)
        x = (rng.random(n) ** 8 * 1e-3).astype(np.float32); x[rng.integers(0, n, 3)] = np.float32(1e10)
        yield "far-out values", x
        yield "tiny", (rng.random(n) * 1e-38).astype(np.float32)                                              # denormals and values below 2^-104
        yield "huge", (rng.random(n) * 1e33).astype(np.float32)
        yield "one value", np.full(n, 3.0, dtype=np.float32)
        x = np.zeros(n, dtype=np.float32); x[n // 2] = 1.0
        yield "a single non-zero", x
        yield "powers of two", np.exp2(rng.integers(-20, 20, n)).astype(np.float32)
        yield "decaying", (np.float32(1e6) / np.arange(1, n + 1, dtype=np.float32) ** 2).astype(np.float32)
        yield "growing", np.arange(1, n + 1, dtype=np.float32) ** 2


@pytest.mark.timeout(900)
def test_device_pick_is_the_sequential_walk(pqv):
    rng = np.random.default_rng(6)
    checked = 0
    for name, md in _cases(rng):
        for workers in (1, 3, 8, 64, 1000):
            if len(md) / min(workers, len(md)) > 57_344:
                continue
            for draw in (0.0, float(rng.random(dtype=np.float32)), 0.99999994):
                total, pick = _reference(md, workers, draw)
                d_total, d_pick, status = _device(pqv, md, workers, draw)
                what = (name, len(md), workers, draw)
                if not (total > 0 and np.isfinite(total)):
                    assert status == 1, what
                    continue
                assert d_total.view(np.uint32) == total.view(np.uint32), what
                if pick is None:
                    assert status == 3, what
                else:
                    assert status == 0 and d_pick == pick, what + (d_pick, pick, status)
                checked += 1
    assert checked > 500


@pytest.mark.timeout(300)
def test_device_pick_hands_bad_values_back(pqv):
    rng = np.random.default_rng(7)
    md = rng.random(10_000, dtype=np.float32)
    for bad in (np.inf, np.nan, -1.0):
        x = md.copy(); x[5000] = bad
        _, _, status = _device(pqv, x, 8, 0.5)
        assert status in (1, 2), (bad, status)
    assert _device(pqv, np.zeros(100, dtype=np.float32), 8, 0.5)[2] == 1        # total == 0: the host draws range_usize (:385)
