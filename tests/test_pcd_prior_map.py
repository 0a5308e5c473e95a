"""The prior-map file of localization mode (adapter/pcd_io.h; reference: utils::readPointCloud, superodom_utils.cpp:16-33 =
pcl::PCDReader, called at laserMapping.cpp:163-173).  PCL is absent: the reader restates the published PCD v0.7 format.  This test
writes the three body encodings from Python -- ascii, binary (array of structures), binary_compressed (LZF stream of the structure
of arrays, with literal runs AND back references) -- with extra fields of other types around x / y / z / intensity, and compares
what adapter/wire_selftest (built by __graft_entry__.build()) reads back, bit for bit."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "adapter", "wire_selftest")


def lzf_compress(data: bytes) -> bytes:
    """Greedy LZF compressor (liblzf format): literal runs of <= 32 bytes, back references of 3 .. 264 bytes within 8 KB."""
    out = bytearray()
    lit = bytearray()
    table = {}
    i, n = 0, len(data)

    def flush():
        nonlocal lit
        while lit:
            run = lit[:32]
            out.append(len(run) - 1)
            out.extend(run)
            lit = lit[32:]

    while i < n:
        key = data[i:i + 3]
        cand = table.get(key) if len(key) == 3 else None
        if len(key) == 3:
            table[key] = i
        if cand is not None and 0 < i - cand <= 8192:
            length = 3
            while i + length < n and length < 264 and data[cand + length] == data[i + length]:
                length += 1
            flush()
            dist = i - cand - 1
            l2 = length - 2
            if l2 < 7:
                out.append((l2 << 5) | (dist >> 8))
            else:
                out.append((7 << 5) | (dist >> 8))
                out.append(l2 - 7)
            out.append(dist & 0xFF)
            i += length
        else:
            lit.append(data[i])
            i += 1
    flush()
    return bytes(out)


def write_pcd(path, encoding, pts, intensity, ring, stamp):
    n = len(pts)
    header = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS ring x y z intensity t\nSIZE 2 4 4 4 4 8\nTYPE U F F F F F\n"
              f"COUNT 1 1 1 1 1 1\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA {encoding}\n").encode()
    with open(path, "wb") as f:
        f.write(header)
        if encoding == "ascii":
            for i in range(n):
                f.write((f"{int(ring[i])} {pts[i, 0]!r} {pts[i, 1]!r} {pts[i, 2]!r} {intensity[i]!r} {stamp[i]!r}\n").replace("np.float32(", "").replace("np.float64(", "").replace(")", "").encode())
        elif encoding == "binary":
            for i in range(n):
                f.write(struct.pack("<Hffffd", int(ring[i]), pts[i, 0], pts[i, 1], pts[i, 2], intensity[i], stamp[i]))
        else:
            soa = ring.astype("<u2").tobytes() + pts[:, 0].astype("<f4").tobytes() + pts[:, 1].astype("<f4").tobytes() + \
                pts[:, 2].astype("<f4").tobytes() + intensity.astype("<f4").tobytes() + stamp.astype("<f8").tobytes()
            comp = lzf_compress(soa)
            assert len(comp) < len(soa), "the test cloud must exercise back references"
            f.write(struct.pack("<II", len(comp), len(soa)) + comp)


@pytest.mark.skipif(not os.path.exists(TOOL), reason="adapter/wire_selftest not built (python __graft_entry__.py)")
@pytest.mark.parametrize("encoding", ["ascii", "binary", "binary_compressed"])
def test_pcd_reader_reads_what_was_written(tmp_path, encoding):
    rng = np.random.default_rng(5)
    n = 3000
    pts = np.round(rng.uniform(-60, 60, (n, 3)), 2).astype(np.float32)  # (two decimals: repeated byte patterns for the compressor)
    pts[::7] = pts[0]  # duplicates -> long back references in the structure of arrays
    pts[11] = [np.nan, 1.0, 2.0]  # pcl::PCDReader keeps non-finite points; the map insert's VoxelGrid drops them later
    intensity = (np.arange(n) % 17).astype(np.float32)
    ring = (np.arange(n) % 128).astype(np.uint16)
    stamp = (1.0e9 + np.arange(n) * 1e-4).astype(np.float64)
    path = str(tmp_path / f"map_{encoding}.pcd")
    out = str(tmp_path / "out.f32")
    write_pcd(path, encoding, pts, intensity, ring, stamp)
    r = subprocess.run([TOOL, "pcd", path, out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert f"points={n}" in r.stdout
    got = np.fromfile(out, np.float32).reshape(-1, 4)
    want = np.c_[pts, intensity].astype(np.float32)
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) or (encoding == "ascii" and np.array_equal(np.isnan(got), np.isnan(want))
                                                                         and np.array_equal(np.nan_to_num(got), np.nan_to_num(want)))


@pytest.mark.skipif(not os.path.exists(TOOL), reason="adapter/wire_selftest not built (python __graft_entry__.py)")
def test_pcd_reader_refuses_what_it_cannot_read(tmp_path):
    out = str(tmp_path / "out.f32")
    r = subprocess.run([TOOL, "pcd", str(tmp_path / "absent.pcd"), out], capture_output=True, text=True)
    assert r.returncode == 3 and "does not exist" in r.stderr  # superodom_utils.cpp:17-21
    bad = tmp_path / "bad.pcd"
    bad.write_bytes(b"VERSION 0.7\nFIELDS a b\nSIZE 4 4\nTYPE F F\nCOUNT 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA binary\n" + b"\0" * 8)
    r = subprocess.run([TOOL, "pcd", str(bad), out], capture_output=True, text=True)
    assert r.returncode == 3 and "x / y / z" in r.stderr
    short = tmp_path / "short.pcd"
    short.write_bytes(b"VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA binary\n" + b"\0" * 20)
    r = subprocess.run([TOOL, "pcd", str(short), out], capture_output=True, text=True)
    assert r.returncode == 3 and "shorter" in r.stderr
    corrupt = tmp_path / "corrupt.pcd"
    corrupt.write_bytes(b"VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 2\nHEIGHT 1\nPOINTS 2\nDATA binary_compressed\n" +
                        struct.pack("<II", 3, 24) + bytes([0xE0, 0x05, 0x00]))
    r = subprocess.run([TOOL, "pcd", str(corrupt), out], capture_output=True, text=True)
    assert r.returncode == 3 and "LZF" in r.stderr
    # hostile / malformed headers: every one of them must come back as "cannot read" -- exit code 3, the path on which
    # the node switches to mapping mode (laserMapping.cpp:165-171) -- never as an uncaught exception (abort, exit code 134)
    head = b"VERSION 0.7\nFIELDS x y z\nSIZE %s\nTYPE F F F\nCOUNT %s\nWIDTH %s\nHEIGHT %s\nPOINTS %s\nDATA %s\n"
    cases = {
        "points_not_a_number": head % (b"4 4 4", b"1 1 1", b"2", b"1", b"two", b"binary") + b"\0" * 24,
        "points_negative": head % (b"4 4 4", b"1 1 1", b"2", b"1", b"-2", b"binary") + b"\0" * 24,
        "points_huge_binary": head % (b"4 4 4", b"1 1 1", b"2", b"1", b"18446744073709551615", b"binary") + b"\0" * 24,
        "points_huge_ascii": head % (b"4 4 4", b"1 1 1", b"2", b"1", b"4000000000", b"ascii") + b"1 2 3\n4 5 6\n",
        "points_huge_compressed": head % (b"4 4 4", b"1 1 1", b"2", b"1", b"4000000000", b"binary_compressed") + struct.pack("<II", 3, 24) + bytes([0, 1, 2]),
        "width_times_height_overflows": (b"VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 4294967296\nHEIGHT 4294967296\nDATA binary\n" + b"\0" * 24),
        "size_not_a_number": head % (b"4 x 4", b"1 1 1", b"2", b"1", b"2", b"binary") + b"\0" * 24,
        "count_out_of_range": head % (b"4 4 4", b"1 99999999999999999999 1", b"2", b"1", b"2", b"binary") + b"\0" * 24,
    }
    for name, blob in cases.items():
        f = tmp_path / (name + ".pcd")
        f.write_bytes(blob)
        r = subprocess.run([TOOL, "pcd", str(f), out], capture_output=True, text=True)
        assert r.returncode == 3 and "PCD" in r.stderr, (name, r.returncode, r.stderr[-200:])
