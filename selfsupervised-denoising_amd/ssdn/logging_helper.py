"""Logging setup of the CLI / trainer (stands where /root/reference/ssdn/ssdn/logging_helper.py:16-88 stands): root logger ->
console + `<run_dir>/log.txt`.  colorlog / colored_traceback are optional dependencies of the reference that are absent here;
plain formatting is used.  `ScalarWriter` stands where the reference's SummaryWriter stands
(train.py:410-438): the same `add_scalar(tag, value, step)` calls land in `<run_dir>/scalars.csv` AND in a TensorBoard event
file `<run_dir>/events.out.tfevents.*` written natively (TFRecord framing + hand-encoded Event / Summary protobufs: the
tensorboard package is absent here; with it installed the file opens in TensorBoard like the reference's)."""
import logging
import os
import socket
import struct
import sys
import time

_FMT = "%(asctime)s %(levelname)-8s %(name)s: %(message)s"
_file_handlers = {}


def setup(log_dir: str = None, filename: str = "log.txt", level=logging.INFO):
    root = logging.getLogger()
    root.setLevel(level)
    if not any(isinstance(h, logging.StreamHandler) and getattr(h, "_ssdn", False) for h in root.handlers):
        h = logging.StreamHandler(sys.stdout)
        h.setFormatter(logging.Formatter(_FMT, "%H:%M:%S"))
        h._ssdn = True
        root.addHandler(h)
    if log_dir is not None:
        os.makedirs(log_dir, exist_ok=True)
        path = os.path.abspath(os.path.join(log_dir, filename))
        if path not in _file_handlers:
            fh = logging.FileHandler(path)
            fh.setFormatter(logging.Formatter(_FMT))
            root.addHandler(fh)
            _file_handlers[path] = fh


# ---- TensorBoard event files without the tensorboard package ---------------------------------------------------------------------
_CRC_TABLE = []


def crc32c(data: bytes) -> int:
    """CRC-32C (Castagnoli), the checksum of the TFRecord framing"""
    if not _CRC_TABLE:
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
            _CRC_TABLE.append(c)
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _len_field(num: int, payload: bytes) -> bytes:
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def encode_event(wall_time: float, step: int = 0, file_version: str = None, tag: str = None, value: float = None) -> bytes:
    """tensorflow.Event { wall_time = 1 (double); step = 2 (int64); file_version = 3 (string) | summary = 5 { value = 1 {
    tag = 1 (string); simple_value = 2 (float) } } }"""
    ev = b"\x09" + struct.pack("<d", wall_time)
    if step:
        ev += b"\x10" + _varint(step)
    if file_version is not None:
        ev += _len_field(3, file_version.encode())
    if tag is not None:
        val = _len_field(1, tag.encode()) + b"\x15" + struct.pack("<f", float(value))
        ev += _len_field(5, _len_field(1, val))
    return ev


def tfrecord(data: bytes) -> bytes:
    head = struct.pack("<Q", len(data))
    return head + struct.pack("<I", _masked_crc(head)) + data + struct.pack("<I", _masked_crc(data))


class EventFileWriter:
    def __init__(self, log_dir: str):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "events.out.tfevents.%010d.%s.%d" % (int(time.time()), socket.gethostname() or "host", os.getpid()))
        self._f = open(self.path, "ab")
        self._f.write(tfrecord(encode_event(time.time(), file_version="brain.Event:2")))
        self._f.flush()

    def add_scalar(self, tag: str, value: float, step: int):
        self._f.write(tfrecord(encode_event(time.time(), int(step), tag=tag, value=value)))
        self._f.flush()

    def close(self):
        self._f.close()


class ScalarWriter:
    def __init__(self, log_dir: str, purge_step: int = None):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "scalars.csv")
        if purge_step is not None and os.path.exists(self.path):       # resuming: drop records from the resume point onwards
            keep = [ln for i, ln in enumerate(open(self.path)) if i == 0 or int(ln.split(",")[1]) < purge_step]
            open(self.path, "w").writelines(keep)
        if not os.path.exists(self.path):
            open(self.path, "w").write("tag,step,value\n")
        self._tb = EventFileWriter(log_dir)     # (a resumed run appends a new event file, like SummaryWriter(purge_step=...))

    def add_scalar(self, tag: str, value, step: int):
        v = float(value)
        with open(self.path, "a") as f:
            f.write("%s,%d,%.9g\n" % (tag, int(step), v))
        if self._tb is not None:
            self._tb.add_scalar(tag, v, step)

    def close(self):
        if self._tb is not None:
            self._tb.close()
