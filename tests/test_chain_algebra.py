"""The arithmetic behind the centroid chains (rescan_amd/csrc/rs_icp_estimate.hip, "grid chains"), restated in numpy
and held against the thing itself: a sequential fp32 sum (lib/rs/icp.h:136-148 adds a million weighted points that way).

  * inside one binade a sequential fp32 sum is an integer sum: s = M u, RN( s + x ) = ( M + rndne( x / u ) ) u;
  * an addend exactly half way between two grid points goes to the EVEN neighbour: what it adds depends on the parity of
    the value it meets — after which that value is even, so a stretch with ties is  M -> M + D + tau[ M & 1 ];
  * such records compose: advances add, the intervals of admissible starts intersect after shifting, taus chain through
    the parity each record starts from.

No GPU, no library: the kernels are held against the replay on the device (tests/test_gpu_parity.py); this pins the
derivation they implement, ties included, on data built to hit them."""
import numpy as np

M_LO, M_HI = 1 << 23, (1 << 24) - 1
F32 = np.float32


def seq_sum(s, xs):
    s = F32(s)
    for x in xs:
        s = F32(s + F32(x))
    return s


def decode(s):
    b = int(np.float32(s).view(np.uint32))
    return (b >> 23) & 255, b >> 31, (b & 0x7FFFFF) | M_LO


def encode(E, sg, M):
    return np.uint32((sg << 31) | (E << 23) | (M & 0x7FFFFF)).view(np.float32)


def record(xs, E, sg):
    """The stretch xs as a function on the mantissas of binade E (sign sg): (lo, hi, D, tau0, tau1), or None.
    k_chain_segrecs' loop: ties to the even neighbour, later ties decided by the first one."""
    P, pmin, pmax, seen, cpar, pfa = 0, 0, 0, False, 0, 0
    for x in xs:
        y = np.ldexp(np.float64(-x if sg else x), 150 - E)          # x / ulp: exact
        if not abs(y) < 2.0 ** 23:
            return None                                             # too big for this binade's grid
        rn = np.rint(y)
        if abs(y - rn) == 0.5:
            kl = int(np.floor(y))
            if not seen:
                seen, cpar = True, (P + kl) & 1
                P += kl; pfa = P                                    # M + P + ((M + cpar) & 1): even from here on
            else:
                P += kl + ((P - pfa + kl) & 1)
        else:
            P += int(rn)
        pmin, pmax = min(pmin, P), max(pmax, P)
    if seen:
        pmax += 1
    lo, hi = max(M_LO, M_LO + 1 - pmin), min(M_HI, M_HI - 1 - pmax)
    if lo > hi:
        return None
    return (lo, hi, P, cpar if seen else 0, (1 - cpar) if seen else 0)


def apply(rec, M):
    lo, hi, D, t0, t1 = rec
    if not (lo <= M and M + max(t0, t1) <= hi):
        return None
    return M + D + (t1 if M & 1 else t0)


def compose(recs):
    """chain_compose_block: one record for a run of records (None if any is, or if the taus outgrow two bits)."""
    if any(r is None for r in recs):
        return None
    ex, lo, hi, t0, t1, tmax = 0, M_LO, M_HI, 0, 0, 0
    for (l, h, D, a0, a1) in recs:
        lo, hi = max(lo, l - ex), min(hi, h - ex)
        if a0 or a1:
            t0, t1 = t0 + (a1 if (ex + t0) & 1 else a0), t1 + (a1 if (1 + ex + t1) & 1 else a0)
            tmax += max(a0, a1)
        ex += D
    hi -= tmax
    if lo > hi or t0 > 3 or t1 > 3:
        return None
    return (lo, hi, ex, t0, t1)


def walk(xs, seg=64, blk=8):
    """The sum of xs by records where they hold the value, addend by addend where they do not; counts both."""
    s, n_rec, n_seq = F32(0.0), 0, 0
    segs = [xs[i:i + seg] for i in range(0, len(xs), seg)]
    i = 0
    while i < len(segs):
        E, sg, M = decode(s)
        if 0 < E < 255 and i % blk == 0 and i + blk <= len(segs):      # a whole block at once?
            r = compose([record(x, E, sg) for x in segs[i:i + blk]])
            m2 = apply(r, M) if r else None
            if m2 is not None:
                s = encode(E, sg, m2); n_rec += 1; i += blk; continue
        r = record(segs[i], E, sg) if 0 < E < 255 else None
        m2 = apply(r, M) if r else None
        if m2 is not None:
            s = encode(E, sg, m2); n_rec += 1
        else:
            s = seq_sum(s, segs[i]); n_seq += 1
        i += 1
    return s, n_rec, n_seq


def test_records_reproduce_the_sequential_sum():
    rng = np.random.default_rng(7)
    for trial in range(6):
        n = 64 * 96
        w = rng.random(n).astype(F32)
        p = (rng.random(n).astype(F32) * F32(6.0) - F32(3.0 if trial % 2 else 0.0))      # a positive and a centred coordinate
        xs = (w * p).astype(F32)
        got, n_rec, n_seq = walk(xs)
        assert got.view(np.uint32) == seq_sum(0.0, xs).view(np.uint32), trial
        assert n_rec > n_seq or trial % 2                                                 # (sums that grow are all but records)


def test_ties_go_to_the_even_neighbour_whatever_the_start():
    """Addends that are odd multiples of half a grid step: every one of them is a tie.  For every start mantissa near the
    ends and in the middle of the interval a record admits, the record's result is the sequential sum's."""
    rng = np.random.default_rng(11)
    E = 140                                                     # ulp 2^-10
    u = np.ldexp(1.0, E - 150)
    for trial in range(40):
        k = rng.integers(-40, 40, 64)
        half = rng.random(64) < (0.5 if trial % 2 else 0.05)
        xs = ((k + 0.5 * half) * u).astype(F32)                 # exact in fp32
        rec = record(xs, E, 0)
        assert rec is not None
        lo, hi = rec[0], rec[1] - max(rec[3], rec[4])
        for M in {lo, lo + 1, hi - 1, hi, (lo + hi) // 2, (lo + hi) // 2 + 1}:
            want = seq_sum(encode(E, 0, M), xs)
            assert apply(rec, M) == decode(want)[2] and decode(want)[0] == E, (trial, M)


def test_composition_is_the_records_in_sequence():
    rng = np.random.default_rng(13)
    E = 138
    u = np.ldexp(1.0, E - 150)
    for trial in range(30):
        recs, xs_all = [], []
        for _ in range(8):
            k = rng.integers(-30, 60, 64)
            half = rng.random(64) < 0.03
            xs = ((k + 0.5 * half) * u).astype(F32)
            recs.append(record(xs, E, 0)); xs_all.append(xs)
        whole = compose(recs)
        if whole is None:
            continue
        lo, hi = whole[0], whole[1] - max(whole[3], whole[4])
        for M in {lo, hi, (lo + hi) // 2, (lo + hi) // 2 + 1}:
            m = M
            for r in recs:
                m = apply(r, m)
                assert m is not None
            assert apply(whole, M) == m
            assert decode(seq_sum(encode(E, 0, M), np.concatenate(xs_all)))[2] == m
