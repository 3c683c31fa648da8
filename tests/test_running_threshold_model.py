"""Model check of the running-threshold counters of wide_filter_kernel (pq-vector_amd/csrc/kernels_screen.hip, "Running
threshold"): per query two 64-bit words of 8-bit counters, counter of bin B = number of appended pairs in bin B or
nearer, word 0 = bins 8..1 and word 1 = bins 12..9 with the NEARER bin in the LOWER byte; an append in bin b adds 1
to the counters of bins 1..b with ONE 64-bit add per word, and nothing stops a byte from wrapping.  The kernel
publishes "bin B holds k pairs" when a byte shows exactly k (2 <= k <= 128).  This test replays random append
sequences -- thousands of appends per query, so bytes wrap many times -- with the kernel's add masks in exact 64-bit
arithmetic and checks the safety claim: whenever a byte shows a value v <= 128, the bin truly holds at least v pairs.
(Plain Python: no GPU, no library.)"""
import random

MASK64 = (1 << 64) - 1
ONES = 0x0101010101010101


def add_masks(b):
    """The kernel's add0 / add1 for an append in bin b (0..12)."""
    add0 = ONES if b >= 8 else ((ONES << (8 * (8 - b))) & MASK64 if b >= 1 else 0)
    add1 = (((ONES & 0xFFFFFFFF) << (8 * (12 - b))) & 0xFFFFFFFF) if b >= 9 else 0
    return add0, add1


def shown(w0, w1):
    """bin -> displayed byte."""
    out = {}
    for j in range(8):
        out[8 - j] = (w0 >> (8 * j)) & 0xFF
    for j in range(4):
        out[12 - j] = (w1 >> (8 * j)) & 0xFF
    return out


def test_add_masks_touch_exactly_bins_1_to_b():
    for b in range(13):
        a0, a1 = add_masks(b)
        s = shown(a0, a1)
        assert all(s[B] == (1 if B <= b else 0) for B in range(1, 13)), (b, s)


def test_wrapping_counters_never_overstate_small_counts():
    rng = random.Random(20240917)
    for trial in range(300):
        n = rng.choice([40, 300, 1500, 6000])
        # bin distributions that make near bins, far bins or everything wrap
        style = trial % 4
        w0 = w1 = 0
        true = [0] * 13
        for _ in range(n):
            if style == 0:
                b = rng.randint(0, 12)
            elif style == 1:
                b = 12 if rng.random() < 0.9 else rng.randint(0, 12)       # almost everything in the nearest bin
            elif style == 2:
                b = rng.choice([1, 1, 1, 2, 9, 12])                         # far bins dominate
            else:
                b = min(12, int(rng.expovariate(0.25)))
            a0, a1 = add_masks(b)
            w0 = (w0 + a0) & MASK64                                          # the byte carries of a 64-bit add
            w1 = (w1 + a1) & MASK64
            for B in range(1, b + 1):
                true[B] += 1
            for B, v in shown(w0, w1).items():
                # the kernel acts on v == k for 2 <= k <= 128 and then needs true[B] >= k
                if v <= 128:
                    assert true[B] >= v, (trial, B, v, true[B])
        # sanity: the sequences do wrap
        if n >= 1500 and style in (0, 1, 3):
            assert max(true) > 255
