"""Height-map sources of the C-ABI (host side: no GPU needed): PNG decoding against PNGs written here with every
scan-line filter, Perlin terrains (determinism, range, smoothness, steps), the text format."""
import struct
import zlib

import numpy as np
import pytest

from raisimlib_amd import RsbError
from raisimlib_amd.world import heightmap_from_png, heightmap_from_text, heightmap_perlin


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def write_png(path, img, depth, color_type, filters, interlace=0, chunk=40):
    """img: [h, w, channels] ints; filters: per-row filter type (cycled).  A tiny independent PNG encoder."""
    h, w, ch = img.shape
    bps = depth // 8
    bpp = ch * bps
    rows = []
    for r in range(h):
        raw = b"".join(int(v).to_bytes(bps, "big") for v in img[r].reshape(-1))
        prev = rows[-1][1] if rows else bytes(len(raw))
        ft = filters[r % len(filters)]
        out = bytearray()
        for i, x in enumerate(raw):
            a = raw[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = [0, a, b, (a + b) // 2, _paeth(a, b, c)][ft]
            out.append((x - pred) & 255)
        rows.append((bytes([ft]) + bytes(out), raw))
    comp = zlib.compress(b"".join(r[0] for r in rows), 6)

    def chunk_(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    data = b"\x89PNG\r\n\x1a\n" + chunk_(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, interlace))
    for i in range(0, len(comp), chunk):                 # several IDAT chunks on purpose
        data += chunk_(b"IDAT", comp[i:i + chunk])
    data += chunk_(b"IEND", b"")
    open(path, "wb").write(data)


@pytest.mark.parametrize("depth,color_type,ch", [(8, 0, 1), (16, 0, 1), (8, 2, 3), (16, 6, 4), (8, 4, 2)])
def test_png_height_map_decodes_exactly(tmp_path, built_lib, depth, color_type, ch):
    rng = np.random.default_rng(depth + color_type)
    h, w = 13, 17
    img = rng.integers(0, 2 ** depth, (h, w, ch))
    p = tmp_path / "hm.png"
    write_png(p, img, depth, color_type, filters=[0, 1, 2, 3, 4])
    got = heightmap_from_png(p, height_scale=2.5, height_offset=-0.3)
    want = (img[:, :, 0] / (2 ** depth - 1) * 2.5 - 0.3).astype(np.float32)
    assert got.shape == (h, w) and np.allclose(got, want, rtol=0, atol=1e-6)


def test_png_errors_are_reported(tmp_path, built_lib):
    p = tmp_path / "x.png"
    p.write_bytes(b"not a png at all, but long enough to pass the size check........")
    with pytest.raises(RsbError, match="not a PNG"):
        heightmap_from_png(p)
    write_png(p, np.zeros((4, 4, 1), int), 8, 0, [0], interlace=1)
    with pytest.raises(RsbError, match="interlaced"):
        heightmap_from_png(p)
    with pytest.raises(RsbError, match="cannot open"):
        heightmap_from_png(tmp_path / "missing.png")


def test_perlin_terrain_properties(built_lib):
    a = heightmap_perlin(128, 96, 12.8, 9.6, frequency=0.4, z_scale=0.5, seed=7)
    b = heightmap_perlin(128, 96, 12.8, 9.6, frequency=0.4, z_scale=0.5, seed=7)
    c = heightmap_perlin(128, 96, 12.8, 9.6, frequency=0.4, z_scale=0.5, seed=8)
    assert a.shape == (96, 128) and np.array_equal(a, b) and not np.allclose(a, c)
    assert np.abs(a).max() < 0.5 * 2.0 and a.std() > 0.02                  # |noise| <~ 1, geometric sum of gains < 2
    assert np.abs(np.diff(a, axis=0)).max() < 0.2 and np.abs(np.diff(a, axis=1)).max() < 0.2     # continuous terrain
    lo = heightmap_perlin(128, 96, 12.8, 9.6, frequency=0.05, z_scale=0.5, seed=7, fractal_octaves=1)
    hi = heightmap_perlin(128, 96, 12.8, 9.6, frequency=0.8, z_scale=0.5, seed=7, fractal_octaves=1)
    assert np.abs(np.diff(lo, axis=1)).mean() < 0.3 * np.abs(np.diff(hi, axis=1)).mean()          # frequency = roughness
    st = heightmap_perlin(64, 64, 6.4, 6.4, frequency=0.4, z_scale=0.5, seed=7, step_size=0.05, height_offset=1.0)
    assert np.allclose(np.round((st - 1.0) / 0.05), (st - 1.0) / 0.05, atol=1e-4) and len(np.unique(st)) > 3


def test_text_height_map_round_trip(tmp_path, built_lib):
    h = np.random.default_rng(0).normal(size=(9, 11)).astype(np.float32)
    p = tmp_path / "hm.txt"
    p.write_text("11 9 5.5 4.5\n" + "\n".join(" ".join(repr(float(v)) for v in row) for row in h))
    got, sx, sy = heightmap_from_text(p)
    assert (sx, sy) == (5.5, 4.5) and np.array_equal(got, h)
    p.write_text("11 9 5.5 4.5\n1 2 3")
    with pytest.raises(RsbError):
        heightmap_from_text(p)
