"""TUM RGB-D on-disk formats: PNG decode (stand-in for cv::imread), association files, trajectory
lines.  The host-only C functions are checked against the pure-Python restatement in oracle/tum.py; the GPU test replays a
synthetic sequence written in TUM layout.  PARITY UNPINNED (no TUM data and no OpenCV in the container)."""
import os
import struct
import zlib

import numpy as np
import pytest


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def write_png(path, samples, bit_depth, color_type, palette=None, filters=None, idat_split=None):
    """Minimal PNG writer for the tests: samples[h][w][channels] ints; every row may use its own filter type (0..4)."""
    samples = np.asarray(samples)
    h, w, ch = samples.shape
    bits = ch * bit_depth
    stride = (w * bits + 7) // 8
    bpp = max(1, bits // 8)
    rows = []
    for y in range(h):
        line = bytearray(stride)
        for x in range(w):
            for k in range(ch):
                s, v = x * ch + k, int(samples[y, x, k])
                if bit_depth == 16:
                    line[2 * s], line[2 * s + 1] = v >> 8, v & 255
                elif bit_depth == 8:
                    line[s] = v
                else:
                    per = 8 // bit_depth
                    line[s // per] |= v << ((per - 1 - s % per) * bit_depth)
        rows.append(line)
    raw = bytearray()
    prev = bytearray(stride)
    for y, line in enumerate(rows):
        f = filters[y % len(filters)] if filters else 0
        raw.append(f)
        for i in range(stride):
            a = line[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            p = [0, a, b, (a + b) // 2, _paeth(a, b, c)][f]
            raw.append((line[i] - p) & 255)
        prev = line

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)

    z = zlib.compress(bytes(raw), 6)
    parts = [z] if not idat_split else [z[:idat_split], z[idat_split:]]
    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, color_type, 0, 0, 0)))
        if palette is not None:
            fh.write(chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes()))
        fh.write(chunk(b"tEXt", b"Comment\x00test"))
        for p in parts:
            fh.write(chunk(b"IDAT", p))
        fh.write(chunk(b"IEND", b""))


@pytest.fixture(scope="module")
def tum():
    from dvo_slam_amd import tum as t

    return t


@pytest.fixture(scope="module")
def otum(orc):
    from oracle import tum as t

    return t


CASES = [  # (color_type, bit_depth, channels)
    (2, 8, 3), (2, 16, 3), (6, 8, 4), (6, 16, 4), (0, 8, 1), (0, 16, 1), (4, 8, 2), (4, 16, 2), (0, 1, 1), (0, 2, 1), (0, 4, 1),
    (3, 8, 1), (3, 4, 1), (3, 2, 1), (3, 1, 1),
]


@pytest.mark.parametrize("color_type,bit_depth,channels", CASES)
def test_png_decode_matches_pure_python_decoder(tmp_path, tum, otum, color_type, bit_depth, channels):
    rng = np.random.default_rng(color_type * 100 + bit_depth)
    w, h = 37, 23  # odd sizes: partial bytes at low bit depths
    samples = rng.integers(0, 1 << bit_depth, size=(h, w, channels))
    samples[5:9] = samples[4]  # flat rows make Up / Paeth interesting
    palette = rng.integers(0, 256, size=(1 << bit_depth, 3)) if color_type == 3 else None
    path = str(tmp_path / "img.png")
    write_png(path, samples, bit_depth, color_type, palette, filters=[0, 1, 2, 3, 4, 4, 3, 1], idat_split=40)
    assert tum.png_info(path) == (w, h, 3 if color_type == 3 else channels, bit_depth)
    got = tum.imread_color(path)
    want = otum.imread_color(path)
    assert got.shape == (h, w, 3) and got.dtype == np.uint8 and np.array_equal(got, want)
    # and against the written samples directly
    if color_type == 2 and bit_depth == 8:
        assert np.array_equal(got, samples[..., ::-1])
    if color_type in (0, 4):
        d = tum.imread_depth(path)
        assert d.dtype == np.uint16 and np.array_equal(d, samples[..., 0]) and np.array_equal(d, otum.imread_depth(path))
    else:
        with pytest.raises(Exception):
            tum.imread_depth(path)


def test_png_reader_rejects_broken_files(tmp_path, tum):
    path = str(tmp_path / "a.png")
    write_png(path, np.zeros((4, 4, 3), int), 8, 2)
    data = bytearray(open(path, "rb").read())
    with pytest.raises(Exception):
        tum.png_info(str(tmp_path / "missing.png"))
    bad = bytearray(data)
    bad[-20] ^= 0x55  # flips a bit inside the last chunks: CRC mismatch
    open(str(tmp_path / "crc.png"), "wb").write(bad)
    with pytest.raises(Exception):
        tum.imread_color(str(tmp_path / "crc.png"))
    open(str(tmp_path / "short.png"), "wb").write(data[:30])
    with pytest.raises(Exception):
        tum.imread_color(str(tmp_path / "short.png"))
    open(str(tmp_path / "notpng.png"), "wb").write(b"P6 4 4 255 " + bytes(48))
    with pytest.raises(Exception):
        tum.png_info(str(tmp_path / "notpng.png"))


def test_png_reader_rejects_absurd_header_dimensions(tmp_path, tum):
    """a crafted IHDR (width / height up to 2^31 - 1) must come back as a format error, never as an overflowing size
    computation or an exception thrown through the C ABI"""
    import struct
    import zlib

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)

    for w, h in ((0x7FFFFFFF, 0x7FFFFFFF), (65536, 65536), (16385, 4), (4, 16385)):
        path = str(tmp_path / f"huge_{w}_{h}.png")
        with open(path, "wb") as fh:
            fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 0, 0, 0, 0)))
            fh.write(chunk(b"IDAT", zlib.compress(b"\x00" * 64)))
            fh.write(chunk(b"IEND", b""))
        with pytest.raises(Exception) as e:
            tum.png_info(path)
        assert getattr(e.value, "status", 12) == 12  # DVO_AMD_ERR_FORMAT
        with pytest.raises(Exception):
            tum.imread_depth(path)


def test_trajectory_line_known_answers_and_oracle(tum, otum, synth):
    # a double holds a 2011 time stamp to ~2.4e-7 s: ros::Time::fromSec turns the parsed 1305031102.175304 into ...175303936 ns
    assert tum.format_trajectory_line(1305031102.175304, np.eye(4)) == "1305031102.175303936 0 0 0 0 0 0 1 \n"
    assert tum.format_trajectory_line(12.5, np.eye(4)) == "12.500000000 0 0 0 0 0 0 1 \n"
    T = np.eye(4)
    T[:3, :3] = [[-1, 0, 0], [0, -1, 0], [0, 0, 1]]  # half turn about z: Eigen's trace <= 0 branch
    T[:3, 3] = [1.5, -0.25, 1e-7]
    assert tum.format_trajectory_line(2.0, T) == "2.000000000 1.5 -0.25 1e-07 0 0 1 0 \n"
    rng = np.random.default_rng(3)
    for k in range(200):
        xi = rng.normal(size=6) * np.array([1, 1, 1, 2.5, 2.5, 2.5])
        xi[3:] *= min(1.0, 3.1 / max(np.linalg.norm(xi[3:]), 1e-9))
        T = synth.se3_exp(xi)
        ts = float(rng.uniform(0, 2e9)) if k % 3 else float(rng.integers(0, 2 ** 31))
        assert tum.format_trajectory_line(ts, T) == otum.trajectory_line(ts, T), (ts, xi)
    assert tum.format_trajectory_line(1.9999999996, np.eye(4)).startswith("2.000000000 ")  # nsec rounds up into sec


def test_association_file_reader(tmp_path, tum):
    p = tmp_path / "assoc.txt"
    p.write_text("# a comment\n# another one\n1.5 rgb/1.5.png 1.52 depth/1.52.png\n"
                 "2.5 rgb/2.5.png 2.51 depth/2.51.png\n3.25 rgb/3.25.png\t3.26   depth/3.26.png\n")
    e = tum.read_assoc(str(p))
    assert [(x.RgbTimestamp, x.RgbFile, x.DepthTimestamp, x.DepthFile) for x in e] == [
        (1.5, "rgb/1.5.png", 1.52, "depth/1.52.png"), (2.5, "rgb/2.5.png", 2.51, "depth/2.51.png"),
        (3.25, "rgb/3.25.png", 3.26, "depth/3.26.png")]
    q = tum.read_assoc(str(p), reference_trailing_entry=True)  # the reference's extra entry after a final newline
    assert len(q) == 4 and (q[3].RgbTimestamp, q[3].RgbFile, q[3].DepthFile) == (0.0, "rgb/3.25.png", "depth/3.26.png")
    p.write_text("1.5 a.png 1.5 b.png")  # no trailing newline: no extra entry in the reference either
    assert len(tum.read_assoc(str(p), reference_trailing_entry=True)) == 1
    g = tmp_path / "gt.txt"
    g.write_text("# ground truth\n1.0 1 2 3 0 0 0 1\n2.0 0 0 0 0 0 0.7071067811865476 0.7071067811865476\n")
    gt = tum.read_groundtruth(str(g))
    assert gt[0][0] == 1.0 and np.allclose(gt[0][1][:3, 3], [1, 2, 3]) and np.allclose(gt[1][1][:2, :2], [[0, -1], [1, 0]])


def _write_sequence(synth, root, n_frames, w, h):
    os.makedirs(os.path.join(root, "rgb"))
    os.makedirs(os.path.join(root, "depth"))
    poses = synth.stream_poses(n_frames)
    lines = ["# synthetic sequence in TUM layout\n"]
    raws = []
    for t, T in enumerate(poses):
        I, Z = synth.render(w, h, T, frame_id=t)
        bgr, raw_z = synth.to_raw(I, Z)
        ts = 1305031102.175304 + t / 30.0
        write_png(os.path.join(root, "rgb", f"{ts:.6f}.png"), bgr[..., ::-1], 8, 2, filters=[1, 2, 4])
        write_png(os.path.join(root, "depth", f"{ts:.6f}.png"), raw_z[..., None], 16, 0, filters=[2, 1])
        lines.append(f"{ts:.6f} rgb/{ts:.6f}.png {ts:.6f} depth/{ts:.6f}.png\n")
        raws.append((bgr, raw_z))
    open(os.path.join(root, "assoc.txt"), "w").writelines(lines)
    return poses, raws


@pytest.mark.gpu
def test_replay_of_a_synthetic_tum_sequence(tmp_path, tum, otum, orc, synth):
    from dvo_slam_amd import capi

    if capi.lib().dvo_amd_device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the MI355X box")
    w, h, n = 320, 240, 4
    root = str(tmp_path / "seq")
    poses, raws = _write_sequence(synth, root, n, w, h)
    K = synth.intrinsics_for(w, h)
    pairs = tum.read_assoc(os.path.join(root, "assoc.txt"))
    assert len(pairs) == n
    # load(): decoded + ingested + pyramid on the GPU == the oracle on the same files
    p = tum.load(K, os.path.join(root, pairs[1].RgbFile), os.path.join(root, pairs[1].DepthFile), 4)
    bgr_o, z_o = otum.imread_color(os.path.join(root, pairs[1].RgbFile)), otum.imread_depth(os.path.join(root, pairs[1].DepthFile))
    assert np.array_equal(bgr_o, raws[1][0]) and np.array_equal(z_o, raws[1][1])
    q = orc.Pyramid(orc.ingest_gray(bgr_o), orc.ingest_depth(z_o), K, 4)
    for level in (0, 2):
        for plane in (0, 1, 3, 5):
            assert np.array_equal(p.plane(level, plane), q.plane(level, plane), equal_nan=True)
    # replay: frame-to-frame odometry, trajectory in TUM format
    out = str(tmp_path / "traj.txt")
    traj = tum.replay(os.path.join(root, "assoc.txt"), out, K=K, config=capi.Config(FirstLevel=3, LastLevel=0))
    assert len(traj) == n
    lines = open(out).read().splitlines(keepends=True)
    assert len(lines) == n and all(line == tum.format_trajectory_line(ts, T) for line, (ts, T) in zip(lines, traj))
    assert lines[0] == "1305031102.175303936 0 0 0 0 0 0 1 \n"
    back = tum.read_groundtruth(out)
    for (ts, T), T_cam in zip(back, poses):
        # camera pose of frame t in frame-0 coordinates is the inverse of the scene-to-camera transform used to render it
        assert synth.pose_error(T, np.linalg.inv(T_cam)) < 5e-3
