"""The arithmetic the MFMA matcher (csrc/match.hip) rests on, restated in numpy and checked against the oracle on the CPU: with
descriptor bits b mapped to int8 1 - 2 b (queries) and 2 b - 1 (trains) the int8 dot product over the 256 bit positions is
2 d - 256, d the Hamming distance; key = (acc << 15) + ((256 << 15) + trainIdx) = d << 16 | trainIdx; the two smallest keys per
query - kept per half of the train rows and merged, with padding rows pushed above every real key - are OpenCV's knnMatch(k = 2)
order (smaller distance first, ties to the lower train index), and Lowe's ratio test on them gives the oracle's matches."""
import numpy as np

import oracle_py as O

INVALID = 0xFF000000


def expand(desc, negate):
    bits = np.unpackbits(desc, axis=1, bitorder="little").astype(np.int32)     # bit position p = byte p / 8, bit p % 8
    v = 1 - 2 * bits
    return -v if negate else v


def knn2_by_gemm(q, t, tile=32):
    nq, nt = len(q), len(t)
    Q, T = expand(q, False), expand(t, True)
    b1 = np.full((2, nq), 0xFFFFFFFF, np.int64)        # [lane half h][query]: rows (g & 3) + 8 (g >> 2) + 4 h of a tile
    b2 = b1.copy()
    for t0 in range(0, nt, tile):
        rows = np.arange(t0, t0 + tile)
        Tt = np.zeros((tile, 256), np.int32)
        ok = rows < nt
        Tt[ok] = T[rows[ok]]                            # padding rows: an all-zero descriptor would be -1 everywhere; any value does
        acc = Tt @ Q.T                                  # [train row][query] = 2 d - 256
        for g in range(16):
            for h in range(2):
                r = (g & 3) + 8 * (g >> 2) + 4 * h
                base = (256 << 15) + t0 + r
                if t0 + r >= nt:
                    base = INVALID
                key = ((acc[r].astype(np.int64) << 15) + base) & 0xFFFFFFFF
                lo = np.minimum(b1[h], key)
                b2[h] = np.median(np.stack([b1[h], b2[h], key]), axis=0).astype(np.int64)   # v_med3_u32 of (b1, b2, key), b1 <= b2
                b1[h] = lo
    m1 = np.minimum(b1[0], b1[1])
    m2 = np.minimum(np.maximum(b1[0], b1[1]), np.minimum(b2[0], b2[1]))
    m1[m1 >= INVALID - (1 << 24)] = 0xFFFFFFFF
    m2[m2 >= INVALID - (1 << 24)] = 0xFFFFFFFF
    return m1, m2


def ratio_matches(m1, m2, ratio):
    out = []
    for qi, (a, b) in enumerate(zip(m1, m2)):
        if a != 0xFFFFFFFF and b != 0xFFFFFFFF and float(a >> 16) < ratio * float(b >> 16):
            out.append((qi, int(a & 0xFFFF), float(a >> 16)))
    return out


def test_pm1_dot_is_hamming_and_keys_are_knn2_order():
    rng = np.random.default_rng(5)
    for nq, nt in ((70, 1), (5, 2), (64, 33), (130, 95), (40, 64)):
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        t[nt // 2] = t[0]                               # exact ties between train rows
        q[0] = t[0]
        q[1] = t[0] ^ np.eye(1, 32, 3, dtype=np.uint8)[0] * 4     # one bit away
        d = (np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2)).astype(np.int64)
        assert np.array_equal(expand(t, True) @ expand(q, False).T, (2 * d - 256).T)
        m1, m2 = knn2_by_gemm(q, t)
        key = (d << 16) | np.arange(nt)[None, :]
        srt = np.sort(key, axis=1)
        assert np.array_equal(m1, srt[:, 0])
        if nt >= 2:
            assert np.array_equal(m2, srt[:, 1])
        else:
            assert (m2 == 0xFFFFFFFF).all()
        for ratio in (0.7, 1.0):
            want = O.match_knn2_ratio(q, t, ratio)
            got = ratio_matches(m1, m2, ratio)
            assert len(got) == len(want)
            assert [g[0] for g in got] == [int(x) for x in want["query_idx"]] and [g[1] for g in got] == [int(x) for x in want["train_idx"]]
            assert [g[2] for g in got] == [float(x) for x in want["distance"]]
