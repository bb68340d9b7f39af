"""CPU tests of the drop-in boundary: libcasim.so loads, exports every symbol include/casim.h declares,
and FAILS LOUDLY (no CPU fallback) when no MI355X is visible."""
import ctypes
import os
import re

import pytest

import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import _abi, _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "casim.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(casim_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_abi.PROTOTYPES)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.casim_abi_version() == _abi.ABI_VERSION


def test_library_is_built_for_gfx950_only():
    blob = open(_ffi.LIB_PATH, "rb").read()
    archs = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert archs == {b"gfx950"}, archs


def test_struct_layouts_match_the_header():
    # sizes computed by hand from include/casim.h (LP64): ints first, then pointers
    assert ctypes.sizeof(_abi.Pegs) == 6 * 4 + 15 * 8                  # (+ zone_polarity, ABI 6; + excl_polarity, ABI 8; + req32, req_unit, ABI 10)
    assert ctypes.sizeof(_abi.Groups) == 8 + 19 * 8 + 3 * 8 + 8 + 8
    assert ctypes.sizeof(_abi.OptionQuery) == 8 + 4 * 4 + 9 * 8      # (+ join_stream, ABI 5)
    assert ctypes.sizeof(_abi.Results) == 10 * 8 + 3 * 8
    assert ctypes.sizeof(_abi.Options) == 48 and ctypes.sizeof(_abi.EncoderOptions) == 32   # (+ chain_last_index + 3 reserved words, ABI 9)


@pytest.mark.skipif(kaa.device_count() > 0, reason="a GPU is visible")
def test_engine_fails_loudly_without_a_gpu():
    with pytest.raises(kaa.NoDeviceError):
        kaa.Context(0)
    assert "no CPU path" in _ffi.last_error() or "no HIP device" in _ffi.last_error()


def test_product_package_never_touches_the_oracle():
    """No import / include / link of anything under oracle/ or tests/emu from the product package
    (the CASIM_HOST_EMU branch of casim_device.h is only ever compiled by tests/emu/Makefile)."""
    pkg = os.path.join(ROOT, "kubernetes_autoscaler_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py"):
                text = open(path).read()
                assert not re.search(r"oracle|libcasim_emu|emu_driver", text), path
            elif f == "Makefile":
                text = open(path).read()
                assert "oracle" not in text and "emu" not in text and "CASIM_HOST_EMU" not in text, path
            elif f.endswith((".h", ".hip", ".cpp")):
                for line in open(path, errors="replace"):
                    if line.lstrip().startswith("#include") and "casim_emu.h" not in line:
                        assert "oracle" not in line and "emu" not in line, (path, line)
    blob = open(_ffi.LIB_PATH, "rb").read()
    assert b"orc_estimate" not in blob and b"casim_emu" not in blob


def test_encoder_term_entry_points_reject_bad_handles_and_mixed_use():
    """casim_enc_pod_add_node_affinity_term / casim_enc_node_term_add_requirement / casim_enc_term_set_namespace_selector /
    casim_enc_term_add_namespace_requirement: bad indices and mixed use come back as CASIM_ERR_INVALID (host-only code)."""
    import ctypes as C
    from kubernetes_autoscaler_amd import _abi
    from kubernetes_autoscaler_amd._ffi import lib
    opts = _abi.EncoderOptions(n_res=2)
    e = lib.casim_enc_create(C.byref(opts))
    assert e
    try:
        req = (C.c_int64 * _abi.MAX_RES)(100, 0)
        s = lib.casim_enc_add_pod_spec(e, b"default", req)
        s2 = lib.casim_enc_add_pod_spec(e, b"default", req)
        vals = (C.c_char_p * 1)(b"v")
        assert lib.casim_enc_pod_add_node_affinity_term(e, 99) < 0
        t = lib.casim_enc_pod_add_node_affinity_term(e, s)
        assert t == 0 and lib.casim_enc_pod_add_node_affinity_term(e, s) == 1
        assert lib.casim_enc_node_term_add_requirement(e, s, 0, 0, b"k", b"In", vals, 1) == 0
        assert lib.casim_enc_node_term_add_requirement(e, s, 1, 1, b"metadata.name", b"In", vals, 1) == 0
        assert lib.casim_enc_node_term_add_requirement(e, s, 2, 0, b"k", b"In", vals, 1) < 0      # no such term
        assert lib.casim_enc_node_term_add_requirement(e, s, 0, 0, b"k", b"In", None, 1) < 0      # values missing
        assert lib.casim_enc_node_term_add_requirement(e, s, 0, 0, b"k", b"In", vals, -1) < 0
        assert lib.casim_enc_pod_add_node_affinity_req(e, s, b"k", b"In", vals, 1) < 0            # one NodeSelector per pod
        assert lib.casim_enc_pod_add_node_affinity_req(e, s2, b"k", b"In", vals, 1) == 0
        assert lib.casim_enc_pod_add_node_affinity_term(e, s2) < 0
        # namespace selectors
        assert lib.casim_enc_term_set_namespace_selector(e, s, 0) < 0                              # the pod has no anti-affinity term
        a = lib.casim_enc_pod_add_anti_affinity_term(e, s, b"kubernetes.io/hostname", None, 0)
        assert a == 0
        assert lib.casim_enc_term_add_namespace_requirement(e, s, a, b"team", b"In", vals, 1) < 0  # selector not set yet
        assert lib.casim_enc_term_set_namespace_selector(e, s, a) == 0
        assert lib.casim_enc_term_add_namespace_requirement(e, s, a, b"team", b"In", vals, 1) == 0
        assert lib.casim_enc_term_add_namespace_requirement(e, s, 7, b"team", b"In", vals, 1) < 0
        assert lib.casim_enc_add_namespace(e, b"default") == 0
        assert lib.casim_enc_namespace_add_label(e, b"team-a", b"team", b"a") == 0
        assert lib.casim_enc_add_namespace(None, b"x") < 0
    finally:
        lib.casim_enc_destroy(e)
