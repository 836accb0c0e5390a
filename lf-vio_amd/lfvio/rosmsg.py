"""ROS 1 wire format of the messages estimator_node.cpp consumes, and a minimal rosbag (format 2.0) reader / writer.

No ROS is needed: a bag recorded where the reference runs — `rosbag record /imu0 /feature_tracker/feature
/feature_tracker/restart /vins_estimator/lfvt_bootstrap` — becomes an LFVT trace (host/replay.h) through
tools/bag_to_lfvt.py.  Message layouts are the published ROS 1 serialization of
  sensor_msgs/Imu          header, orientation (4 f64) + covariance (9), angular_velocity (3) + covariance (9),
                           linear_acceleration (3) + covariance (9)                (imu_callback, estimator_node.cpp:136-161)
  sensor_msgs/PointCloud   header, points[] (u32 n, n x 3 f32), channels[] (u32 m, each: string name, f32[] values)
                           (feature_tracker_node.cpp:113-178 -> estimator_node.cpp:292-312)
  std_msgs/Bool            u8                                                        (restart_callback, :187-204)
  std_msgs/Float64MultiArray  layout (dim[], data_offset), f64[] data — the bootstrap record the dump hook of
                           INTEGRATION.md section 3 publishes (247 doubles + the stamp of Headers[WINDOW_SIZE])
header = u32 seq, u32 stamp.sec, u32 stamp.nsec, string frame_id; string = u32 length + bytes; little-endian throughout.
The writer exists for the tests (a synthetic bag with the layout of a real one); the reader handles uncompressed and bz2
chunks (lz4 chunks need `rosbag decompress` first).
"""
import bz2
import struct

import numpy as np


# ---------------------------------------------------------------- messages
def _string(b):
    return struct.pack("<I", len(b)) + b


def _header(seq, stamp, frame_id=b""):
    sec = int(np.floor(stamp))
    nsec = int(round((stamp - sec) * 1e9))
    if nsec >= 1000000000:
        sec, nsec = sec + 1, nsec - 1000000000
    return struct.pack("<III", seq, sec, nsec) + _string(frame_id)


def _read_header(buf, o):
    seq, sec, nsec = struct.unpack_from("<III", buf, o)
    (n,) = struct.unpack_from("<I", buf, o + 12)
    return sec + nsec * 1e-9, o + 16 + n


def ser_imu(seq, stamp, acc, gyr, frame_id=b"imu"):
    z9 = struct.pack("<9d", *([0.0] * 9))
    return (_header(seq, stamp, frame_id) + struct.pack("<4d", 0.0, 0.0, 0.0, 1.0) + z9 + struct.pack("<3d", *gyr) + z9
            + struct.pack("<3d", *acc) + z9)


def de_imu(buf):
    """-> (stamp, acc[3], gyr[3])  (header.stamp.toSec(), linear_acceleration, angular_velocity)"""
    stamp, o = _read_header(buf, 0)
    o += 8 * (4 + 9)
    gyr = struct.unpack_from("<3d", buf, o)
    o += 8 * (3 + 9)
    acc = struct.unpack_from("<3d", buf, o)
    return stamp, np.array(acc), np.array(gyr)


CHANNELS = (b"id_of_point", b"u_of_point", b"v_of_point", b"velocity_x_of_point", b"velocity_y_of_point", b"velocity_z_of_point")


def ser_pointcloud(seq, stamp, rec9, frame_id=b"world"):
    """rec9: [n, 9] float32 as an LFVT feature record holds them: point x y z, then the six channel values."""
    a = np.asarray(rec9, dtype="<f4").reshape(-1, 9)
    out = _header(seq, stamp, frame_id) + struct.pack("<I", len(a)) + a[:, 0:3].tobytes() + struct.pack("<I", 6)
    for c, name in enumerate(CHANNELS):
        out += _string(name) + struct.pack("<I", len(a)) + np.ascontiguousarray(a[:, 3 + c]).tobytes()
    return out


def de_pointcloud(buf):
    """-> (stamp, [n, 9] float32)"""
    stamp, o = _read_header(buf, 0)
    (n,) = struct.unpack_from("<I", buf, o)
    o += 4
    pts = np.frombuffer(buf, dtype="<f4", count=3 * n, offset=o).reshape(n, 3)
    o += 12 * n
    (m,) = struct.unpack_from("<I", buf, o)
    o += 4
    rec = np.zeros((n, 9), dtype="<f4")
    rec[:, 0:3] = pts
    for c in range(m):
        (ln,) = struct.unpack_from("<I", buf, o)
        o += 4 + ln
        (k,) = struct.unpack_from("<I", buf, o)
        o += 4
        if c < 6:
            assert k == n, "channel length differs from the number of points"
            rec[:, 3 + c] = np.frombuffer(buf, dtype="<f4", count=k, offset=o)
        o += 4 * k
    assert m >= 6, "the tracker publishes six channels (feature_tracker_node.cpp:152-163)"
    return stamp, rec


def ser_bool(v):
    return struct.pack("<B", 1 if v else 0)


def de_bool(buf):
    return buf[0] != 0


def ser_f64_array(data):
    d = np.asarray(data, dtype="<f8").ravel()
    return struct.pack("<I", 0) + struct.pack("<I", 0) + struct.pack("<I", len(d)) + d.tobytes()


def de_f64_array(buf):
    (ndim,) = struct.unpack_from("<I", buf, 0)
    o = 4
    for _ in range(ndim):
        (ln,) = struct.unpack_from("<I", buf, o)
        o += 4 + ln + 8
    o += 4  # data_offset
    (n,) = struct.unpack_from("<I", buf, o)
    return np.frombuffer(buf, dtype="<f8", count=n, offset=o + 4).copy()


TYPES = {
    "sensor_msgs/Imu": "6a62c6daae103f4ff57a132d6f95cec2",
    "sensor_msgs/PointCloud": "d8e9c3f5afbdd8a130fd1d2763945fca",
    "std_msgs/Bool": "8b94c1b53db61fb6aed406028ad6332a",
    "std_msgs/Float64MultiArray": "4b7d974086d4060e7db4613a7e6c3ba4",
}


# ---------------------------------------------------------------- bag container (format 2.0)
def _fields(d):
    out = b""
    for k, v in d.items():
        f = k.encode() + b"=" + v
        out += struct.pack("<I", len(f)) + f
    return out


def _record(header, data):
    h = _fields(header)
    return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data


def _time(t):
    sec = int(np.floor(t))
    return struct.pack("<II", sec, int(round((t - sec) * 1e9)) % 1000000000)


class BagWriter:
    """Enough of a bag for the reader below and for `rosbag info / play`: bag header, chunks with connection + message
    records, index-less (a real recorder also writes index and chunk-info records, which readers may skip)."""

    def __init__(self, path, compression="none", chunk_messages=64):
        self.f = open(path, "wb")
        self.f.write(b"#ROSBAG V2.0\n")
        fields = {"op": b"\x03", "index_pos": struct.pack("<Q", 0), "conn_count": struct.pack("<I", 0), "chunk_count": struct.pack("<I", 0)}
        self.f.write(_record(fields, b" " * (4096 - 8 - len(_fields(fields)))))  # the bag header record is padded to 4096 bytes
        self.compression, self.per_chunk = compression, chunk_messages
        self.conns, self.buf, self.n = {}, b"", 0

    def _conn(self, topic, mtype):
        if topic in self.conns:
            return self.conns[topic]
        cid = len(self.conns)
        self.conns[topic] = cid
        data = _fields({"topic": topic.encode(), "type": mtype.encode(), "md5sum": TYPES[mtype].encode(), "message_definition": b""})
        self.buf += _record({"op": b"\x07", "conn": struct.pack("<I", cid), "topic": topic.encode()}, data)
        return cid

    def write(self, topic, mtype, t, payload):
        cid = self._conn(topic, mtype)
        self.buf += _record({"op": b"\x02", "conn": struct.pack("<I", cid), "time": _time(t)}, payload)
        self.n += 1
        if self.n % self.per_chunk == 0:
            self._flush()

    def _flush(self):
        if not self.buf:
            return
        data = bz2.compress(self.buf) if self.compression == "bz2" else self.buf
        self.f.write(_record({"op": b"\x05", "compression": self.compression.encode(), "size": struct.pack("<I", len(self.buf))}, data))
        self.buf = b""

    def close(self):
        self._flush()
        self.f.close()


def _parse_fields(h):
    out, o = {}, 0
    while o < len(h):
        (n,) = struct.unpack_from("<I", h, o)
        k, _, v = h[o + 4:o + 4 + n].partition(b"=")
        out[k.decode()] = v
        o += 4 + n
    return out


def _records(buf, o=0):
    while o + 4 <= len(buf):
        (hl,) = struct.unpack_from("<I", buf, o)
        h = _parse_fields(buf[o + 4:o + 4 + hl])
        (dl,) = struct.unpack_from("<I", buf, o + 4 + hl)
        yield h, buf[o + 8 + hl:o + 8 + hl + dl]
        o += 8 + hl + dl


def read_bag(path):
    """Yield (topic, type, receive time, payload bytes) in file order (the order the messages were received in)."""
    raw = open(path, "rb").read()
    assert raw.startswith(b"#ROSBAG V2.0\n"), "not a rosbag 2.0 file"
    conns = {}

    def walk(buf, o=0):
        for h, data in _records(buf, o):
            op = h["op"][0]
            if op == 0x03:  # bag header (its data is padding)
                continue
            if op == 0x05:
                comp = h["compression"].decode()
                if comp == "bz2":
                    data = bz2.decompress(data)
                elif comp != "none":
                    raise ValueError(f"chunk compression {comp}: run `rosbag decompress` first")
                yield from walk(data)
            elif op == 0x07:
                cid = struct.unpack("<I", h["conn"])[0]
                f = _parse_fields(data)
                conns[cid] = (h["topic"].decode(), f["type"].decode())
            elif op == 0x02:
                cid = struct.unpack("<I", h["conn"])[0]
                sec, nsec = struct.unpack("<II", h["time"])
                topic, mtype = conns[cid]
                yield topic, mtype, sec + nsec * 1e-9, data
            # 0x04 index data, 0x06 chunk info: not needed to read the messages in order

    yield from walk(raw, 13)
