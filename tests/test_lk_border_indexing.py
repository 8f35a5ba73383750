"""Host-side restatement of lk_border_kernel's thread -> dword enumeration (csrc/lk.hip): every byte of the reflect-101 border the
LK tile loads may touch - 32 rows above and below, 32 columns left, at least 32 right of the image - is written, no dword twice,
nothing outside the plane row, and the values are numpy's reflect padding (cv::BORDER_REFLECT_101)."""
import numpy as np
import pytest

PAD, PADR = 32, 48


def enumerate_dwords(w, h):
    pd = (PAD + w + PADR) >> 2
    xr = w & ~3
    nr = (w + PADR - xr) >> 2
    band = 2 * PAD * pd
    total = band + h * (PAD // 4 + nr)
    out = []
    for i in range(total):
        if i < band:
            r = i // pd
            x0 = -PAD + 4 * (i - r * pd)
            y = r - PAD if r < PAD else h + (r - PAD)
        else:
            per = PAD // 4 + nr
            j = i - band
            y = j // per
            c = j - y * per
            x0 = -PAD + 4 * c if c < PAD // 4 else xr + 4 * (c - PAD // 4)
        out.append((x0, y))
    return out


def refl(p, n):
    while p < 0 or p >= n:
        p = -p if p < 0 else 2 * (n - 1) - p
    return p


@pytest.mark.parametrize("w,h", [(640, 360), (321, 182), (161, 91), (23, 22), (203, 179)])
def test_border_dwords_cover_the_border_once(w, h):
    pitch = (w + PAD + PADR + 63) // 64 * 64
    rng = np.random.default_rng(w + h)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    plane = np.full((h + 2 * PAD, pitch), -1, np.int32)
    plane[PAD:PAD + h, PAD:PAD + w] = img
    seen = set()
    for x0, y in enumerate_dwords(w, h):
        assert (x0, y) not in seen and x0 % 4 == 0 and -PAD <= x0 and x0 + 4 <= pitch - PAD and -PAD <= y < h + PAD
        seen.add((x0, y))
        for b in range(4):
            v = img[refl(y, h), refl(x0 + b, w)]
            if 0 <= y < h and 0 <= x0 + b < w:
                assert plane[PAD + y, PAD + x0 + b] == v          # an in-image byte of a straddling dword gets its own value back
            plane[PAD + y, PAD + x0 + b] = v
    want = np.pad(img, ((PAD, PAD), (PAD, PAD)), mode="reflect") if min(w, h) > PAD else None
    got = plane[:, :w + 2 * PAD]
    assert (got >= 0).all()                                        # everything the tile loads may read (to column w + 31) is written
    if want is not None:
        assert np.array_equal(got, want)
