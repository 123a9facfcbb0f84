"""Minimal ROS1 bag (format v2.0) reader for the three message types the front-end consumes, and the converter to the flat
log of replay.py / tools/replay_log.cpp ("ROS-bag-v2 or flat-log reader").  No ROS installation needed.

Bag v2.0 layout: "#ROSBAG V2.0\\n", then records  <u32 header_len><header fields><u32 data_len><data>; a header field is
<u32 len>name=value.  op 0x03 bag header, 0x05 chunk (compression none / bz2 / lz4 = the LZ4 frame format roslz4 writes), 0x07 connection
(topic, type), 0x02 message data (conn, time), 0x04 / 0x06 index records (skipped).  Messages are ROS1-serialised
(little-endian; string = u32 len + bytes; T[] = u32 count + items; Header = u32 seq, u32 secs, u32 nsecs, string frame_id):
  sensor_msgs/Imu        Header, Quaternion orientation, f64[9], Vector3 angular_velocity, f64[9], Vector3 linear_acceleration, f64[9]
  nav_msgs/Odometry      Header, string child_frame_id, Pose (Point, Quaternion xyzw), f64[36], Twist (2 x Vector3), f64[36]
  sensor_msgs/LaserScan  Header, f32 angle_min, angle_max, angle_increment, time_increment, scan_time, range_min, range_max,
                         f32[] ranges, f32[] intensities
Conversions follow the reference's sensor constructors (src/trajectory/sensor.h:25-29, :103-121): header stamps, the odometry
quaternion normalised and turned into a rotation matrix with Eigen's formula.
"""
import bz2
import struct

import numpy as np


def lz4_block_decode(src, out=None):
    """One LZ4 block (the sequence format: token = literal length << 4 | match length - 4, 255-continued lengths, literals, u16
    little-endian match offset; the last sequence ends after its literals).  Matches may overlap their own output (offset < length:
    run-length encoding), so they are copied in pieces of at most `offset` bytes.  `out`: bytearray holding the history the block may
    reference (linked blocks); the decoded bytes are appended to it."""
    out = bytearray() if out is None else out
    i, n = 0, len(src)
    while i < n:
        tok = src[i]
        i += 1
        ll = tok >> 4
        if ll == 15:
            while True:
                b = src[i]
                i += 1
                ll += b
                if b != 255:
                    break
        if i + ll > n:
            raise ValueError("lz4: literal run past the end of the block")
        out += src[i:i + ll]
        i += ll
        if i >= n:                       # the last sequence has no match part
            break
        off = src[i] | (src[i + 1] << 8)
        i += 2
        if off == 0 or off > len(out):
            raise ValueError("lz4: match offset outside the decoded data")
        ml = tok & 15
        if ml == 15:
            while True:
                b = src[i]
                i += 1
                ml += b
                if b != 255:
                    break
        ml += 4
        start = len(out) - off
        while ml > 0:
            piece = out[start:start + min(ml, off)]
            out += piece
            ml -= len(piece)
            start += len(piece)
    return out


def lz4_frame_decode(buf):
    """LZ4 frame (magic 0x184D2204; what roslz4 / `rosbag compress --lz4` writes into a chunk): frame descriptor FLG (version 01, block
    independence, block checksum, content size, content checksum, dictionary id), BD, optional u64 content size / u32 dictionary id, header
    checksum byte; then blocks <u32 size, bit 31 = stored uncompressed> ... until a zero size word; optional xxh32 content checksum
    (verified when the xxhash module is importable).  Several frames may follow each other."""
    out, o = bytearray(), 0
    while o < len(buf):
        (magic,) = struct.unpack_from("<I", buf, o)
        o += 4
        if 0x184D2A50 <= magic <= 0x184D2A5F:              # skippable frame
            (sl,) = struct.unpack_from("<I", buf, o)
            o += 4 + sl
            continue
        if magic != 0x184D2204:
            raise ValueError("lz4: bad frame magic %#x" % magic)
        flg, bd = buf[o], buf[o + 1]
        o += 2
        if (flg >> 6) != 1:
            raise ValueError("lz4: unsupported frame version")
        indep, bsum, csize, csum, dictid = (flg >> 5) & 1, (flg >> 4) & 1, (flg >> 3) & 1, (flg >> 2) & 1, flg & 1
        want = None
        if csize:
            (want,) = struct.unpack_from("<Q", buf, o)
            o += 8
        if dictid:
            o += 4
        o += 1                                             # header checksum byte
        start = len(out)
        while True:
            (bs,) = struct.unpack_from("<I", buf, o)
            o += 4
            if bs == 0:
                break
            raw, bs = bs >> 31, bs & 0x7fffffff
            blk = buf[o:o + bs]
            if len(blk) != bs:
                raise ValueError("lz4: truncated block")
            o += bs + (4 if bsum else 0)
            if raw:
                out += blk
            elif indep:
                out += lz4_block_decode(blk)
            else:                                          # linked blocks: matches reach back into the previous blocks of the frame
                tail = bytearray(out[max(start, len(out) - 65536):])
                k = len(tail)
                out += lz4_block_decode(blk, tail)[k:]
        if csum:
            (cs,) = struct.unpack_from("<I", buf, o)
            o += 4
            try:
                import xxhash
                if xxhash.xxh32(bytes(out[start:]), seed=0).intdigest() != cs:
                    raise ValueError("lz4: content checksum mismatch")
            except ImportError:
                pass
        if want is not None and len(out) - start != want:
            raise ValueError("lz4: content size mismatch")
    return bytes(out)


def _fields(buf):
    out, o = {}, 0
    while o < len(buf):
        (n,) = struct.unpack_from("<I", buf, o)
        o += 4
        name, _, val = buf[o:o + n].partition(b"=")
        out[name.decode()] = val
        o += n
    return out


def _records(buf, o=0, end=None):
    end = len(buf) if end is None else end
    while o + 4 <= end:
        (hl,) = struct.unpack_from("<I", buf, o)
        o += 4
        hdr = _fields(buf[o:o + hl])
        o += hl
        (dl,) = struct.unpack_from("<I", buf, o)
        o += 4
        yield hdr, buf[o:o + dl]
        o += dl


def _header(buf, o):
    seq, secs, nsecs, fl = struct.unpack_from("<IIII", buf, o)
    o += 16 + fl
    return secs + nsecs * 1e-9, o


def parse_imu(buf):
    t, o = _header(buf, 0)
    o += 8 * 4 + 8 * 9                                   # orientation + covariance
    gyro = struct.unpack_from("<3d", buf, o)
    o += 8 * 3 + 8 * 9
    acc = struct.unpack_from("<3d", buf, o)
    return dict(type=0, time=t, acc=np.array(acc), gyro=np.array(gyro))


def parse_odometry(buf):
    t, o = _header(buf, 0)
    (cl,) = struct.unpack_from("<I", buf, o)
    o += 4 + cl
    px, py, pz, qx, qy, qz, qw = struct.unpack_from("<7d", buf, o)
    q = np.array([qw, qx, qy, qz])
    q = q / np.linalg.norm(q)                           # q.normalize()
    w, x, y, z = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z                    # Eigen::Quaternion::toRotationMatrix
    twx, twy, twz, txx, txy, txz, tyy, tyz, tzz = tx * w, ty * w, tz * w, tx * x, ty * x, tz * x, ty * y, tz * y, tz * z
    R = np.array([[1 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1 - (txx + tzz), tyz - twx], [txz - twy, tyz + twx, 1 - (txx + tyy)]])
    return dict(type=1, time=t, R=R, t=np.array([px, py, pz]))


def parse_laserscan(buf):
    t, o = _header(buf, 0)
    amin, amax, ainc, tinc, scan_time, rmin, rmax = struct.unpack_from("<7f", buf, o)
    o += 28
    (n,) = struct.unpack_from("<I", buf, o)
    o += 4
    ranges = np.frombuffer(buf, dtype="<f4", count=n, offset=o).copy()
    return dict(type=3, time=t, angle_min=np.float32(amin), angle_increment=np.float32(ainc), time_increment=np.float32(tinc), ranges=ranges)


PARSERS = {"sensor_msgs/Imu": parse_imu, "nav_msgs/Odometry": parse_odometry, "sensor_msgs/LaserScan": parse_laserscan}


def read_bag(path, topics=None):
    """-> list of message dicts (replay.py layout) of the supported types, in bag order; topics: optional set of names."""
    raw = open(path, "rb").read()
    if not raw.startswith(b"#ROSBAG V2.0\n"):
        raise ValueError("not a ROS bag v2.0 file")
    conns, out = {}, []

    def handle(hdr, data):
        op = hdr["op"][0]
        if op == 0x07:
            c = struct.unpack("<I", hdr["conn"])[0]
            ch = _fields(data)
            conns[c] = (hdr["topic"].decode(), ch.get("type", b"").decode())
        elif op == 0x02:
            c = struct.unpack("<I", hdr["conn"])[0]
            topic, typ = conns.get(c, (None, None))
            if typ in PARSERS and (topics is None or topic in topics):
                m = PARSERS[typ](data)
                m["topic"] = topic
                out.append(m)
        elif op == 0x05:
            comp = hdr["compression"].decode()
            if comp == "bz2":
                data = bz2.decompress(data)
            elif comp == "lz4":
                data = lz4_frame_decode(data)
            elif comp != "none":
                raise ValueError("chunk compression %r is not supported (re-compress the bag: rosbag decompress)" % comp)
            if "size" in hdr and len(hdr["size"]) == 4 and struct.unpack("<I", hdr["size"])[0] != len(data):
                raise ValueError("chunk: %d bytes after decompression, header says %d" % (len(data), struct.unpack("<I", hdr["size"])[0]))
            for h2, d2 in _records(data):
                handle(h2, d2)
    for hdr, data in _records(raw, len(b"#ROSBAG V2.0\n")):
        handle(hdr, data)
    return out


def bag_to_flatlog(bag_path, out_path, imu_topic="/d400/imu0", odom_topic="/odom", scan_topic="/scan"):
    """Topic names default to reference config/office.yaml:1-3.  Messages are written sorted by header stamp."""
    from .replay import write_log
    msgs = read_bag(bag_path, {imu_topic, odom_topic, scan_topic})
    msgs.sort(key=lambda m: m["time"])
    write_log(out_path, msgs)
    return len(msgs)
