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
        data = bz2.compress(body) if compression == "bz2" else body
        out += _record([("op", b"\x05"), ("compression", compression.encode()), ("size", struct.pack("<I", len(body)))], data)
        out += _record([("op", b"\x04"), ("ver", struct.pack("<I", 1)), ("conn", struct.pack("<I", 0)), ("count", struct.pack("<I", 0))], b"")   # index record: skipped
    open(path, "wb").write(out)


@pytest.mark.parametrize("compression", ["none", "bz2"])
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
