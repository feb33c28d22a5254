"""ctypes binding of libkmm_io.so (include/kmm_io.h): read-file bytes straight into the caller's buffer — BGZF
members inflated in parallel at their final place, a gzip stream by a read-ahead thread, plain files by parallel
pread.  Built in-tree with g++ by build(); gz_io / reads_io use it when it is there (KMM_IO_PYTHON=1: the pure-Python
readers instead)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
SO_PATH = os.path.join(_HERE, "libkmm_io.so")
SRC = os.path.join(_HERE, "csrc", "kmm_io.cpp")
DEPS = [SRC, os.path.join(_HERE, "csrc", "kmm_inflate.hpp"), os.path.join(ROOT, "include", "kmm_io.h")]


def _src_mtime():
    return max(os.path.getmtime(d) for d in DEPS if os.path.exists(d))

_c = ctypes
_P = ctypes.c_void_p
SIGNATURES = {
    "kmm_io_open": (_P, [_c.c_char_p, _c.c_int]),
    "kmm_io_kind": (_c.c_int, [_P]),
    "kmm_io_read": (_c.c_int64, [_P, _P, _c.c_int64]),
    "kmm_io_seek": (_c.c_int, [_P, _c.c_int64]),
    "kmm_io_close": (None, [_P]),
    "kmm_io_engine": (_c.c_int, []),
    "kmm_io_error": (_c.c_char_p, []),
}


def build(force=False):
    """g++ -O3 -shared (zlib linked; libdeflate, when installed, is found at run time through dlopen)."""
    if not force and os.path.exists(SO_PATH) and os.path.getmtime(SO_PATH) >= _src_mtime():
        return SO_PATH
    subprocess.check_call([os.environ.get("CXX", "g++"), "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
                           "-I" + os.path.join(ROOT, "include"), "-o", SO_PATH, SRC, "-lz", "-ldl"])
    return SO_PATH


_lib = None


def _fresh():
    """libkmm_io.so exists and is not older than its sources — kmm_io.cpp, the inflater's header, the C header —
    (rebuilt here when it is and a compiler is at hand: an edit of any of them must not leave a stale reader in use)."""
    if not os.path.exists(SO_PATH):
        return False
    if _lib is None and os.path.exists(SRC) and os.path.getmtime(SO_PATH) < _src_mtime():
        try:
            build()
        except FileNotFoundError:       # no compiler on this box: the existing library is what there is
            pass
        except subprocess.CalledProcessError as exc:     # a compile error must not hide behind the old library
            import logging
            logging.getLogger(__name__).error("rebuilding %s failed (%s): the existing, OLDER library stays in use", SO_PATH, exc)
    return True


def available():
    return os.environ.get("KMM_IO_PYTHON") != "1" and _fresh()


def lib():
    global _lib
    if _lib is None:
        _fresh()
        L = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


_threads = None


def affinity_count():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_quota():
    """CPUs' worth of time the cgroup grants (cpu.max: quota / period, rounded up), or None when it sets no limit."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max" and int(period) > 0:
            return max(1, -(-int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return None


def cpu_budget():
    """Cores this process may keep busy: its affinity mask, cut by the cgroup's CPU quota (cpu.max) where there is one —
    a container that sees 256 CPUs with a quota of 16 gets 16 cores' worth of time, and more busy threads are throttled."""
    n, q = affinity_count(), cpu_quota()
    return max(1, n if q is None else min(n, q))


def set_default_threads(n):
    """`kmer_mapper map -t N` (command_line_interface.py:168): worker threads of the readers opened from here on."""
    global _threads
    _threads = max(1, int(n)) if n else None


def default_threads():
    if _threads is not None:
        return _threads
    return max(1, min(16, cpu_budget()))          # the reference CLI's -t default


class NativeStream:
    """Forward-only byte stream over kmm_io_*: read() / readinto() / seek() (plain files) / close()."""

    def __init__(self, path, n_threads=None):
        self._h = lib().kmm_io_open(os.fsencode(str(path)), int(n_threads or default_threads()))
        if not self._h:
            raise OSError(lib().kmm_io_error().decode("utf-8", "replace"))
        self.kind = lib().kmm_io_kind(self._h)          # 0 plain, 1 BGZF, 2 gzip stream

    def readinto(self, b):
        mv = memoryview(b).cast("B")
        n = len(mv)
        if n == 0:
            return 0
        buf = (ctypes.c_uint8 * n).from_buffer(mv)
        done = 0
        while done < n:                                  # (BGZF hands out whole members: fill the buffer)
            got = lib().kmm_io_read(self._h, ctypes.byref(buf, done), n - done)
            if got < 0:
                msg = lib().kmm_io_error().decode("utf-8", "replace")
                raise (EOFError if "ended before" in msg or "truncated" in msg else ValueError)(msg)
            if got == 0:
                break
            done += got
        return done

    def read(self, n=-1):
        if n is None or n < 0:
            parts = []
            while True:
                piece = self.read(1 << 24)
                if not piece:
                    return b"".join(parts)
                parts.append(piece)
        buf = bytearray(n)
        got = self.readinto(buf)
        return bytes(buf[:got])

    def seek(self, pos):
        if lib().kmm_io_seek(self._h, int(pos)) != 0:
            raise OSError(lib().kmm_io_error().decode("utf-8", "replace"))

    def close(self):
        if self._h:
            lib().kmm_io_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
