"""ROS bag v2.0 reader + flat-log converter (2dliw-slam_amd/rosbag_reader.py) on a bag written by a minimal bag writer in
this test (no ROS available here): chunked, with and without bz2 compression, three connections, interleaved index records."""
import bz2
import importlib
import struct

import numpy as np
import pytest


def _field(name, val):
    b = name.encode() + b"=" + val
    return struct.pack("<I", len(b)) + b


def _record(fields, data):
    h = b"".join(_field(k, v) for k, v in fields)
    return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data


def _hdr(seq, t, frame):
    secs = int(t)
    nsecs = int(round((t - secs) * 1e9))
    return struct.pack("<III", seq, secs, nsecs) + struct.pack("<I", len(frame)) + frame


def ser_imu(seq, m):
    return (_hdr(seq, m["time"], b"imu") + struct.pack("<4d", 0, 0, 0, 1) + b"\0" * 72 + struct.pack("<3d", *m["gyro"]) + b"\0" * 72 +
            struct.pack("<3d", *m["acc"]) + b"\0" * 72)


def ser_odom(seq, m, scale=1.0):
    R = np.asarray(m["R"])
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    q = np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w]) * scale   # xyzw, deliberately unnormalised
    return (_hdr(seq, m["time"], b"odom") + struct.pack("<I", 9) + b"base_link" + struct.pack("<3d", *m["t"]) + struct.pack("<4d", *q) + b"\0" * 288 +
            b"\0" * 48 + b"\0" * 288)


def ser_scan(seq, m):
    r = np.asarray(m["ranges"], dtype="<f4")
    return (_hdr(seq, m["time"], b"laser") + struct.pack("<7f", m["angle_min"], -float(m["angle_min"]), m["angle_increment"], m["time_increment"], 0.04, 0.1, 30.0) +
            struct.pack("<I", len(r)) + r.tobytes() + struct.pack("<I", 0))


def lz4_block_encode(data):
    """Greedy LZ4 block encoder for the tests (4-byte hash chain, overlapping matches allowed): written from the block format
    description, independent of the decoder under test."""
    n, i, anchor, out, table = len(data), 0, 0, bytearray(), {}

    def emit(lit, mlen, off):
        ll, ml = len(lit), (mlen - 4 if mlen else 0)
        out.append((min(ll, 15) << 4) | (min(ml, 15) if mlen else 0))
        if ll >= 15:
            r = ll - 15
            while r >= 255:
                out.append(255); r -= 255
            out.append(r)
        out.extend(lit)
        if mlen:
            out.extend(struct.pack("<H", off))
            if ml >= 15:
                r = ml - 15
                while r >= 255:
                    out.append(255); r -= 255
                out.append(r)
    while i + 4 <= n - 5:                  # the last 5 bytes are literals (end-of-block rule)
        key = data[i:i + 4]
        j = table.get(key)
        table[key] = i
        if j is not None and i - j <= 65535:
            m = 4
            while i + m < n - 5 and data[j + m] == data[i + m]:
                m += 1
            emit(data[anchor:i], m, i - j)
            i += m
            anchor = i
        else:
            i += 1
    emit(data[anchor:], 0, 0)
    return bytes(out)


def lz4_frame_encode(data, block=1 << 16, content_checksum=True, content_size=False, store_some=True):
    import xxhash
    flg = (1 << 6) | (1 << 5) | ((1 << 3) if content_size else 0) | ((1 << 2) if content_checksum else 0)
    desc = bytes([flg, 4 << 4]) + (struct.pack("<Q", len(data)) if content_size else b"")
    out = struct.pack("<I", 0x184D2204) + desc + bytes([(xxhash.xxh32(desc).intdigest() >> 8) & 255])
    for k, o in enumerate(range(0, len(data), block)):
        raw = data[o:o + block]
        enc = lz4_block_encode(raw)
        if len(enc) >= len(raw) or (store_some and k % 3 == 2):   # incompressible (or every third) block: stored
            out += struct.pack("<I", len(raw) | 0x80000000) + raw
        else:
            out += struct.pack("<I", len(enc)) + enc
    out += struct.pack("<I", 0)
    if content_checksum:
        out += struct.pack("<I", xxhash.xxh32(data).intdigest())
    return out


def write_bag(path, msgs, compression, chunk=50):
    conn = {0: (0, "/d400/imu0", "sensor_msgs/Imu"), 1: (1, "/odom", "nav_msgs/Odometry"), 3: (2, "/scan", "sensor_msgs/LaserScan")}
    ser = {0: ser_imu, 1: lambda s, m: ser_odom(s, m, 1.7), 3: ser_scan}
    out = b"#ROSBAG V2.0\n"
    out += _record([("op", b"\x03"), ("index_pos", struct.pack("<Q", 0)), ("conn_count", struct.pack("<I", 3)), ("chunk_count", struct.pack("<I", 0))], b" " * 64)
    seen = set()
    for c0 in range(0, len(msgs), chunk):
        body = b""
        for k, m in enumerate(msgs[c0:c0 + chunk]):
            cid, topic, typ = conn[m["type"]]
            if cid not in seen:
                seen.add(cid)
                body += _record([("op", b"\x07"), ("conn", struct.pack("<I", cid)), ("topic", topic.encode())],
                                _field("topic", topic.encode()) + _field("type", typ.encode()) + _field("md5sum", b"0" * 32) + _field("message_definition", b"..."))
            secs = int(m["time"])
            body += _record([("op", b"\x02"), ("conn", struct.pack("<I", cid)), ("time", struct.pack("<II", secs, int((m["time"] - secs) * 1e9)))], ser[m["type"]](c0 + k, m))
        data = bz2.compress(body) if compression == "bz2" else (lz4_frame_encode(body, block=4096, content_size=(c0 // chunk) % 2 == 1) if compression == "lz4" else body)
        out += _record([("op", b"\x05"), ("compression", compression.encode()), ("size", struct.pack("<I", len(body)))], data)
        out += _record([("op", b"\x04"), ("ver", struct.pack("<I", 1)), ("conn", struct.pack("<I", 0)), ("count", struct.pack("<I", 0))], b"")   # index record: skipped
    open(path, "wb").write(out)


def test_lz4_block_and_frame_decoder(liw):
    """Known-answer vectors written by hand from the LZ4 block format, then the greedy test encoder on data with long runs (overlapping
    matches), repeats at distance > 255 and random bytes (stored blocks); linked-block frames; corrupt input is reported."""
    rb = importlib.import_module("2dliw-slam_amd.rosbag_reader")
    # token 0x50: 5 literals, no match (last sequence)
    assert bytes(rb.lz4_block_decode(b"\x50hello")) == b"hello"
    # "abcd" then a match of 8 at offset 4 (overlap: abcdabcd), then 5 literals
    assert bytes(rb.lz4_block_decode(b"\x44abcd\x04\x00\x50vwxyz")) == b"abcdabcdabcdvwxyz"
    # run-length: 1 literal, match offset 1 of length 4 + 15 + 3 = 22 (extended match length)
    assert bytes(rb.lz4_block_decode(b"\x1fa\x01\x00\x03\x50.....")) == b"a" * 23 + b"....."
    # extended literal length: 15 + 2 = 17 literals
    assert bytes(rb.lz4_block_decode(b"\xf0\x02" + b"0123456789abcdefg")) == b"0123456789abcdefg"
    rng = np.random.default_rng(5)
    blobs = [b"", b"x", b"short", bytes(1000), b"abc" * 700, rng.integers(0, 256, 9000, dtype=np.uint8).tobytes(),
             (b"A" * 300 + rng.integers(0, 4, 500, dtype=np.uint8).tobytes()) * 40]
    for blob in blobs:
        assert bytes(rb.lz4_block_decode(lz4_block_encode(blob))) == blob
        for kw in (dict(), dict(content_size=True), dict(content_checksum=False, block=777)):
            assert rb.lz4_frame_decode(lz4_frame_encode(blob, **kw)) == blob
    # two frames back to back + a skippable frame in between
    two = lz4_frame_encode(b"first " * 50) + struct.pack("<II", 0x184D2A50, 3) + b"xyz" + lz4_frame_encode(b"second " * 50)
    assert rb.lz4_frame_decode(two) == b"first " * 50 + b"second " * 50
    # linked blocks (block-independence flag clear): the second block's match reaches into the first
    import xxhash
    desc = bytes([1 << 6, 4 << 4])
    b1, b2 = b"\x80abcdefgh", b"\x04\x08\x00\x30xyz"            # 8 literals | match of 8 at offset 8 (into block 1), then 3 literals
    linked = struct.pack("<I", 0x184D2204) + desc + bytes([(xxhash.xxh32(desc).intdigest() >> 8) & 255]) + struct.pack("<I", len(b1)) + b1 + struct.pack("<I", len(b2)) + b2 + struct.pack("<I", 0)
    assert rb.lz4_frame_decode(linked) == b"abcdefghabcdefghxyz"
    bad = bytearray(lz4_frame_encode(b"payload " * 100, store_some=False))
    bad[-1] ^= 1                                                   # content checksum
    with pytest.raises(ValueError):
        rb.lz4_frame_decode(bytes(bad))
    with pytest.raises(ValueError):
        rb.lz4_block_decode(b"\x04\x09\x00")                      # match offset before the start of the data
    with pytest.raises(ValueError):
        rb.lz4_frame_decode(b"\x00\x01\x02\x03rest")


@pytest.mark.parametrize("compression", ["none", "bz2", "lz4"])
def test_bag_round_trip(liw, synth, tmp_path, compression):
    replay = importlib.import_module("2dliw-slam_amd.replay")
    rb = importlib.import_module("2dliw-slam_amd.rosbag_reader")
    msgs, _ = replay.make_log(synth.office_params(), duration=1.5, seed=4, n_rays=180)
    # bag order is not time order: shuffle within a window, the converter sorts by header stamp
    rng = np.random.default_rng(0)
    shuffled = list(msgs)
    for i in range(0, len(shuffled) - 5, 5):
        j = i + int(rng.integers(0, 5))
        shuffled[i], shuffled[j] = shuffled[j], shuffled[i]
    bag = str(tmp_path / "t.bag")
    write_bag(bag, shuffled, compression)
    got = rb.read_bag(bag)
    assert len(got) == len(msgs) and {m["topic"] for m in got} == {"/d400/imu0", "/odom", "/scan"}
    n = rb.bag_to_flatlog(bag, str(tmp_path / "flat.bin"))
    assert n == len(msgs)
    got.sort(key=lambda m: m["time"])
    for a, b in zip(got, msgs):
        assert a["type"] == b["type"] and abs(a["time"] - b["time"]) < 2e-9
        if a["type"] == 0:
            assert np.array_equal(a["acc"], b["acc"]) and np.array_equal(a["gyro"], b["gyro"])
        elif a["type"] == 1:
            assert np.array_equal(a["t"], b["t"]) and np.abs(a["R"] - b["R"]).max() < 1e-12      # quaternion normalised on the way in
        else:
            assert np.array_equal(a["ranges"], b["ranges"]) and a["angle_increment"] == b["angle_increment"] and a["time_increment"] == b["time_increment"]
    # the flat file is what replay.write_log produces from the parsed messages: same record count and size
    replay.write_log(str(tmp_path / "ref.bin"), got)
    assert open(str(tmp_path / "flat.bin"), "rb").read() == open(str(tmp_path / "ref.bin"), "rb").read()
    with pytest.raises(ValueError):
        open(str(tmp_path / "bad.bag"), "wb").write(b"not a bag")
        rb.read_bag(str(tmp_path / "bad.bag"))
