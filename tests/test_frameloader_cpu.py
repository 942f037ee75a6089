"""Next row f2, CPU part: the image decoder behind the re-stated FrameLoader (PNG via zlib, binary
PGM/PPM) against images written here with Python's zlib -- every PNG colour type / filter the loader may meet."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "monocular_slam_amd", "host", "mslam_imageio_check")


def write_png(path, arr, bit_depth=8, palette=None, filters="cycle"):
    """arr: (h, w) or (h, w, c) integer samples in file channel order (RGB[A] / gray[+alpha]); palette: (n, 3) -> colour type 3."""
    a = arr if arr.ndim == 3 else arr[:, :, None]
    h, w, c = a.shape
    ctype = 3 if palette is not None else {1: 0, 2: 4, 3: 2, 4: 6}[c]
    if bit_depth == 16:
        rows = a.astype(">u2").reshape(h, -1).view(np.uint8)
    elif bit_depth == 8:
        rows = a.astype(np.uint8).reshape(h, -1)
    else:  # packed sub-byte gray / palette indices
        per = 8 // bit_depth
        pad = (-w) % per
        v = np.pad(a[:, :, 0].astype(np.uint8), ((0, 0), (0, pad)))
        rows = np.zeros((h, v.shape[1] // per), np.uint8)
        for k in range(per):
            rows |= v[:, k::per] << (8 - bit_depth * (k + 1))
    bpp = max(1, c * bit_depth // 8) if palette is None else 1
    raw = bytearray()
    prev = np.zeros(rows.shape[1], np.int32)
    for y in range(h):
        cur = rows[y].astype(np.int32)
        ft = y % 5 if filters == "cycle" else int(filters)
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0:
            out = cur
        elif ft == 1:
            out = cur - left
        elif ft == 2:
            out = cur - prev
        elif ft == 3:
            out = cur - ((left + prev) >> 1)
        else:
            p = left + prev - upleft
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            out = cur - pred
        raw.append(ft)
        raw += (out & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    z = zlib.compress(bytes(raw), 6)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, ctype, 0, 0, 0)))
        if palette is not None:
            f.write(chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes()))
        half = len(z) // 2  # two IDAT chunks: the stream may be split anywhere
        f.write(chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b""))


def decode(tmp, name, *flags):
    out = tmp / (name + ".raw")
    r = subprocess.run([CHECK, str(tmp / name), str(out), *flags], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    rows, cols, ch, typ = map(int, r.stdout.split())
    return np.fromfile(out, np.float32 if "--depth" in flags else np.uint8).reshape(rows, cols, ch if "--depth" not in flags else 1)


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(CHECK):
        subprocess.run(["make", "-C", os.path.dirname(CHECK), "mslam_imageio_check"], check=True, capture_output=True)


def test_png_colour_types_and_filters(tmp_path):
    rng = np.random.default_rng(1)
    g = rng.integers(0, 256, (37, 53), dtype=np.uint16)
    write_png(tmp_path / "g8.png", g)
    np.testing.assert_array_equal(decode(tmp_path, "g8.png")[:, :, 0], g)
    rgb = rng.integers(0, 256, (21, 34, 3), dtype=np.uint16)
    write_png(tmp_path / "rgb8.png", rgb)
    np.testing.assert_array_equal(decode(tmp_path, "rgb8.png"), rgb[:, :, ::-1])          # imread returns BGR
    rgba = rng.integers(0, 256, (9, 17, 4), dtype=np.uint16)
    write_png(tmp_path / "rgba8.png", rgba)
    np.testing.assert_array_equal(decode(tmp_path, "rgba8.png"), rgba[:, :, [2, 1, 0, 3]])
    ga = rng.integers(0, 256, (9, 17, 2), dtype=np.uint16)
    write_png(tmp_path / "ga8.png", ga)
    np.testing.assert_array_equal(decode(tmp_path, "ga8.png")[:, :, 0], ga[:, :, 0])
    rgb16 = rng.integers(0, 65536, (8, 11, 3), dtype=np.uint32)
    write_png(tmp_path / "rgb16.png", rgb16, bit_depth=16)
    np.testing.assert_array_equal(decode(tmp_path, "rgb16.png"), (rgb16 >> 8)[:, :, ::-1])
    pal = rng.integers(0, 256, (16, 3))
    idx = rng.integers(0, 16, (13, 19), dtype=np.uint16)
    write_png(tmp_path / "pal4.png", idx, bit_depth=4, palette=pal)
    np.testing.assert_array_equal(decode(tmp_path, "pal4.png"), pal[idx][:, :, ::-1])
    bits = rng.integers(0, 2, (10, 21), dtype=np.uint16)
    write_png(tmp_path / "g1.png", bits, bit_depth=1)
    np.testing.assert_array_equal(decode(tmp_path, "g1.png")[:, :, 0], bits * 255)
    for ft in range(5):
        write_png(tmp_path / f"f{ft}.png", rgb, filters=str(ft))
        np.testing.assert_array_equal(decode(tmp_path, f"f{ft}.png"), rgb[:, :, ::-1])


def test_depth_png_and_pnm_and_writer_round_trip(tmp_path):
    rng = np.random.default_rng(2)
    d = rng.integers(0, 65536, (24, 32), dtype=np.uint32)
    write_png(tmp_path / "d16.png", d, bit_depth=16)
    np.testing.assert_array_equal(decode(tmp_path, "d16.png", "--depth")[:, :, 0], d.astype(np.float32))   # TUM depth maps
    g = rng.integers(0, 256, (12, 20), dtype=np.uint8)
    with open(tmp_path / "a.pgm", "wb") as f:
        f.write(b"P5\n# comment\n20 12\n255\n" + g.tobytes())
    np.testing.assert_array_equal(decode(tmp_path, "a.pgm")[:, :, 0], g)
    c = rng.integers(0, 256, (6, 5, 3), dtype=np.uint8)
    with open(tmp_path / "a.ppm", "wb") as f:
        f.write(b"P6 5 6 255\n" + c.tobytes())
    np.testing.assert_array_equal(decode(tmp_path, "a.ppm"), c[:, :, ::-1])
    # the C++ writer's PNG decodes back to the same pixels
    decode(tmp_path, "a.ppm", "--no", str(tmp_path / "b.png"))
    np.testing.assert_array_equal(decode(tmp_path, "b.png"), c[:, :, ::-1])


def test_bad_files_fail_loudly(tmp_path):
    (tmp_path / "x.png").write_bytes(b"not a png at all, definitely not")
    r = subprocess.run([CHECK, str(tmp_path / "x.png"), str(tmp_path / "o.raw")], capture_output=True, text=True)
    assert r.returncode == 1 and "PNG" in r.stderr
    (tmp_path / "x.jpg").write_bytes(b"\xff\xd8\xff")
    r = subprocess.run([CHECK, str(tmp_path / "x.jpg"), str(tmp_path / "o.raw")], capture_output=True, text=True)
    assert r.returncode == 1 and "unsupported" in r.stderr


def test_hostile_headers_are_refused_before_allocating(tmp_path):
    """Header dimensions come from the file: absurd sizes, overflowing digit runs and zero sizes fail through the
    false/err contract instead of allocating gigabytes or overflowing."""
    import struct, zlib

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 0x7FFFFFFF, 0x7FFFFFFF, 8, 0, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(b"\0\0")) + chunk(b"IEND", b"")
    (tmp_path / "huge.png").write_bytes(png)
    r = subprocess.run([CHECK, str(tmp_path / "huge.png"), str(tmp_path / "o.raw")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "out of range" in r.stderr
    for name, body in (("digits.pgm", b"P5 99999999999999999999 2 255\n\0\0"), ("zero.pgm", b"P5 0 4 255\n"), ("maxval.pgm", b"P5 2 2 0\n\0\0\0\0"),
                       ("big.pgm", b"P5 20000 20000 255\n\0")):
        (tmp_path / name).write_bytes(body)
        r = subprocess.run([CHECK, str(tmp_path / name), str(tmp_path / "o.raw")], capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and r.stderr.strip(), name
