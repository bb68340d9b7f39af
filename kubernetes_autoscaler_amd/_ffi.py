"""Loads the product library libcasim.so (HIP kernels for gfx950 + host encoder) via ctypes.

There is deliberately no fallback: if the library is missing the import fails loudly, and every
engine call fails with CASIM_ERR_NO_DEVICE when no MI355X is visible."""
import ctypes
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CASIM_LIB_PATH") or os.path.join(_HERE, "libcasim.so")   # override: profiling builds only


class CasimError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libcasim error {code}: {message}")
        self.code = code
        self.message = message


class NoDeviceError(CasimError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C kubernetes_autoscaler_amd/csrc` (hipcc --offload-arch=gfx950). "
            "kubernetes_autoscaler_amd has no pure-Python / CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)  # RTLD_LOCAL: the library keeps its symbols to itself
    _abi.bind(lib)
    if lib.casim_abi_version() != _abi.ABI_VERSION:
        raise ImportError(f"libcasim ABI {lib.casim_abi_version()} != expected {_abi.ABI_VERSION}: rebuild")
    return lib


lib = _load()


def last_error():
    msg = lib.casim_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, what=""):
    if rc == _abi.OK:
        return
    msg = last_error() or what
    if rc == _abi.ERR_NO_DEVICE:
        raise NoDeviceError(rc, msg)
    raise CasimError(rc, f"{what}: {msg}" if what else msg)
