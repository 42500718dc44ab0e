"""CPU restatement of the TUM file formats around the tracker -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this module.  PARITY UNPINNED.
An independent pure-Python PNG decoder (the PNG specification's inflate + filter rules; stands for cv::imread,
benchmark_slam.cpp:50-51, OpenCV itself being absent) and the trajectory line of benchmark_slam.cpp:490-504 with
ros::Time's and Eigen::Quaterniond's published conversions.  Slow loops: for small test images only.
"""
from __future__ import annotations

import math
import struct
import zlib

import numpy as np

_CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def png_decode(path):
    """-> (samples[h][w][channels] as nested numpy array of ints (palette already expanded to RGB), bit_depth, color_type)"""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, palette = 8, b"", None
    while pos < len(data):
        (length,) = struct.unpack(">I", data[pos:pos + 4])
        ctype = data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + length]
        (crc,) = struct.unpack(">I", data[pos + 8 + length:pos + 12 + length])
        assert crc == zlib.crc32(ctype + body) & 0xFFFFFFFF
        if ctype == b"IHDR":
            w, h, bd, ct, comp, flt, inter = struct.unpack(">IIBBBBB", body)
            assert comp == 0 and flt == 0 and inter == 0
        elif ctype == b"PLTE":
            palette = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif ctype == b"IDAT":
            idat += body
        elif ctype == b"IEND":
            break
        pos += 12 + length
    ch = _CHANNELS[ct]
    bits = ch * bd
    stride = (w * bits + 7) // 8
    bpp = max(1, bits // 8)
    raw = zlib.decompress(idat)
    rows = []
    prev = bytearray(stride)
    for y in range(h):
        f = raw[y * (stride + 1)]
        line = bytearray(raw[y * (stride + 1) + 1:(y + 1) * (stride + 1)])
        for i in range(stride):
            a = line[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if f == 0:
                p = 0
            elif f == 1:
                p = a
            elif f == 2:
                p = b
            elif f == 3:
                p = (a + b) // 2
            else:
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            line[i] = (line[i] + p) & 0xFF
        rows.append(line)
        prev = line
    out = np.zeros((h, w, ch), np.int64)
    for y in range(h):
        line = rows[y]
        for x in range(w):
            for k in range(ch):
                s = x * ch + k
                if bd == 16:
                    v = (line[2 * s] << 8) | line[2 * s + 1]
                elif bd == 8:
                    v = line[s]
                else:
                    per = 8 // bd
                    v = (line[s // per] >> ((per - 1 - s % per) * bd)) & ((1 << bd) - 1)
                out[y, x, k] = v
    if ct == 3:
        out = palette[out[..., 0]].astype(np.int64)
    return out, bd, ct


def imread_color(path):
    """cv::imread(path, 1): 8-bit BGR; 16-bit samples keep the high byte, low bit depths are scaled to 0..255."""
    s, bd, ct = png_decode(path)
    if ct == 3:
        rgb = s
    else:
        if bd == 16:
            s = s >> 8
        elif bd < 8:
            s = s * 255 // ((1 << bd) - 1)
        rgb = np.repeat(s[..., :1], 3, axis=2) if ct in (0, 4) else s[..., :3]
    return rgb[..., ::-1].astype(np.uint8)


def imread_depth(path):
    s, bd, ct = png_decode(path)
    assert ct in (0, 4)
    return s[..., 0].astype(np.uint16)


def ros_time_string(t):
    """ros::Time().fromSec(t) printed with operator<< : sec.nsec with nsec zero-padded to 9 digits."""
    sec = int(math.floor(t))
    nsec = int(math.floor((t - sec) * 1e9 + 0.5))
    if nsec >= 1000000000:
        sec, nsec = sec + 1, nsec - 1000000000
    return "%d.%09d" % (sec, nsec)


def eigen_quaternion(R):
    """Eigen::Quaterniond(R) -> (x, y, z, w)."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    q = [0.0] * 4
    if t > 0:
        t = math.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0], q[1], q[2] = (R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (R[k, j] - R[j, k]) * t
        q[j] = (R[j, i] + R[i, j]) * t
        q[k] = (R[k, i] + R[i, k]) * t
    return q


def trajectory_line(timestamp, T):
    """benchmark_slam.cpp:490-504: default ostream formatting of doubles is %g with 6 significant digits."""
    T = np.asarray(T, dtype=np.float64)
    q = eigen_quaternion(T[:3, :3])
    vals = [T[0, 3], T[1, 3], T[2, 3]] + q
    return ros_time_string(timestamp) + " " + " ".join("%g" % v for v in vals) + " \n"
