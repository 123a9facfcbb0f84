"""On-disk outputs of the front-end over the C ABI of include/liw_io.h: the TUM trajectory file
(`TumWriter`, reference src/trajectory/trajectory.cpp:59-67,549-559) and the `Record` timing tables
(reference src/utilies/record.h:13-126)."""
import ctypes as C

import numpy as np

IO_EXPORTS = ["liw_tum_pose", "liw_tum_format_line", "liw_tum_open", "liw_tum_append", "liw_tum_close", "liw_record_create", "liw_record_destroy",
              "liw_record_begin", "liw_record_end", "liw_record_add_time", "liw_record_add", "liw_record_format", "liw_record_write"]


def _lib():
    from . import lib
    L = lib()
    if not getattr(L, "_io_ready", False):
        L.liw_tum_open.restype = C.c_void_p
        L.liw_record_create.restype = C.c_void_p
        L.liw_record_end.restype = C.c_uint64
        for name in ("liw_tum_close", "liw_record_destroy", "liw_record_begin"):
            getattr(L, name).argtypes = [C.c_void_p]
        L._io_ready = True
    return L


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def tum_pose(prm, p, q):
    """x y z qx qy qz qw of make_tf(p, q) * T_imu_to_wheel"""
    from . import params_struct
    ps = params_struct(prm)
    a, b, out = np.ascontiguousarray(p, dtype=np.float64), np.ascontiguousarray(q, dtype=np.float64), np.zeros(7)
    _lib().liw_tum_pose(C.byref(ps), _pd(a), _pd(b), _pd(out))
    return out


def tum_line(time, pose7):
    buf = C.create_string_buffer(400)
    v = np.ascontiguousarray(pose7, dtype=np.float64)
    n = _lib().liw_tum_format_line(C.c_double(time), _pd(v), buf, C.c_int(400))
    return buf.raw[:n].decode()


class TumWriter:
    def __init__(self, path, prm):
        from . import params_struct
        self._ps = params_struct(prm)
        self.h = C.c_void_p(_lib().liw_tum_open(str(path).encode(), C.byref(self._ps)))
        if not self.h:
            raise OSError("cannot open %s" % path)

    def append(self, time, p, q):
        a, b = np.ascontiguousarray(p, dtype=np.float64), np.ascontiguousarray(q, dtype=np.float64)
        return _lib().liw_tum_append(self.h, C.c_double(time), _pd(a), _pd(b))

    def close(self):
        if self.h:
            _lib().liw_tum_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Record:
    def __init__(self):
        self.h = C.c_void_p(_lib().liw_record_create())

    def begin_record(self):
        _lib().liw_record_begin(self.h)

    def end_record(self, type_name):
        return int(_lib().liw_record_end(self.h, type_name.encode()))

    def add_time(self, type_name, us):
        _lib().liw_record_add_time(self.h, type_name.encode(), C.c_uint64(int(us)))

    def add_record(self, type_name, v):
        _lib().liw_record_add(self.h, type_name.encode(), C.c_uint64(int(v)))

    def format(self):
        n = _lib().liw_record_format(self.h, None, C.c_int(0))
        buf = C.create_string_buffer(n + 1)
        _lib().liw_record_format(self.h, buf, C.c_int(n + 1))
        return buf.raw[:n].decode()

    def write(self, path):
        return _lib().liw_record_write(self.h, str(path).encode())

    def __del__(self):
        try:
            if self.h:
                _lib().liw_record_destroy(self.h)
        except Exception:
            pass
