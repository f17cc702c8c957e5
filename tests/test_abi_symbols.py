"""CPU-only checks of the drop-in boundary: the headers are plain C, and the built library exports every
function they declare (plus the symbols the reference's dispatch macro looks up by name,
pytorch_npu_helper.hpp:224-294,727-748).  No compute call is made here."""
import ctypes
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(ROOT, "cann-ops-adv-vllm_amd", "lib", "libcust_opapi.so")

DECL = re.compile(r"^\s*(?:SPFA_API\s+)?(?:const\s+)?[A-Za-z_][\w\s\*]*?\b((?:acl|spfa|Init|UnInit|Release)\w*)\s*\(", re.M)


def declared_functions():
    names = set()
    for h in sorted(glob.glob(os.path.join(INCLUDE, "*.h"))):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        for m in DECL.finditer(text):
            names.add(m.group(1))
    return sorted(names)


def test_headers_declare_the_reference_entry_points():
    names = declared_functions()
    for op in ("ComputeCent", "SelectPosition", "CentSelect", "SparsePagedAttention", "SparsePagedFusionAttention"):
        assert f"aclnn{op}GetWorkspaceSize" in names
        assert f"aclnn{op}" in names
    for fn in ("aclCreateTensor", "aclDestroyTensor", "aclCreateTensorList", "aclDestroyTensorList",
               "aclCreateIntArray", "aclDestroyIntArray", "aclSetAclOpExecutorRepeatable",
               "aclDestroyAclOpExecutor", "aclGetRecentErrMsg"):
        assert fn in names


@pytest.mark.parametrize("header", sorted(os.path.basename(h) for h in glob.glob(os.path.join(INCLUDE, "*.h"))))
def test_header_is_plain_c(header):
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                        "-I", INCLUDE, os.path.join(INCLUDE, header)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "build the library first (make -C cann-ops-adv-vllm_amd/csrc)"
    lib = ctypes.CDLL(LIB)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    # looked up by name by the reference's dispatch macro, null-tolerated there
    for n in ("InitHugeMemThreadLocal", "UnInitHugeMemThreadLocal", "ReleaseHugeMem"):
        assert hasattr(lib, n), n


def test_phase_one_validates_without_a_gpu():
    """Phase 1 is synchronous host code: null descriptors are refused with the documented status and text."""
    lib = ctypes.CDLL(LIB)
    lib.aclGetRecentErrMsg.restype = ctypes.c_char_p
    ws = ctypes.c_uint64(0)
    ex = ctypes.c_void_p()
    st = lib.aclnnComputeCentGetWorkspaceSize(None, None, None, ctypes.byref(ws), ctypes.byref(ex))
    assert st == 161001
    assert lib.aclGetRecentErrMsg()
    dims = (ctypes.c_int64 * 3)(2, 8, 64)
    strides = (ctypes.c_int64 * 3)(512, 64, 1)
    lib.aclCreateTensor.restype = ctypes.c_void_p
    lib.aclCreateTensor.argtypes = [ctypes.POINTER(ctypes.c_int64), ctypes.c_uint64, ctypes.c_int,
                                    ctypes.POINTER(ctypes.c_int64), ctypes.c_int64, ctypes.c_int,
                                    ctypes.POINTER(ctypes.c_int64), ctypes.c_uint64, ctypes.c_void_p]
    lib.aclDestroyTensor.argtypes = [ctypes.c_void_p]
    # float32 query: rejected as an invalid parameter before anything touches a device
    q = lib.aclCreateTensor(dims, 3, 0, strides, 0, 2, dims, 3, ctypes.c_void_p(0x1000))
    lib.aclnnComputeCentGetWorkspaceSize.argtypes = [ctypes.c_void_p] * 3 + [ctypes.POINTER(ctypes.c_uint64),
                                                                              ctypes.POINTER(ctypes.c_void_p)]
    st = lib.aclnnComputeCentGetWorkspaceSize(q, q, q, ctypes.byref(ws), ctypes.byref(ex))
    assert st == 161002
    lib.aclDestroyTensor(q)
