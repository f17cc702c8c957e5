"""ctypes binding of libcust_opapi.so (the aclnn-style two-phase C-ABI declared in include/*.h).

This is the MI355X counterpart of the reference's extension/csrc/add_custom.cpp helpers
(ConvertTensorToAcl :30-76, ConvertTensorListToAcl :79-102, ConvertToIntArray :105-118) and of its
GetWorkspaceSize -> workspace -> exec -> destroy sequence (:836-929).  PyTorch only supplies device
memory and the current HIP stream; there is no CPU fallback: if the library is missing the import
fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int64, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPFA_LIB", os.path.join(_HERE, "..", "lib", "libcust_opapi.so"))

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"custom_ops: {LIB_PATH} not found. Build it with `make -C cann-ops-adv-vllm_amd/csrc` "
        "(or __graft_entry__.build()); there is no fallback implementation.")

lib = ctypes.CDLL(os.path.abspath(LIB_PATH))

# aclDataType values (include/acl_shim.h)
_ACL_DTYPE = {
    torch.float32: 0, torch.float16: 1, torch.int8: 2, torch.int32: 3, torch.uint8: 4,
    torch.int16: 6, torch.int64: 9, torch.float64: 11, torch.bool: 12, torch.bfloat16: 27,
}
ACL_FORMAT_ND = 2

T = c_void_p  # opaque descriptor pointers


def _sig(name, restype, argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


aclCreateTensor = _sig("aclCreateTensor", T, [POINTER(c_int64), c_uint64, c_int, POINTER(c_int64), c_int64, c_int,
                                              POINTER(c_int64), c_uint64, c_void_p])
aclDestroyTensor = _sig("aclDestroyTensor", c_int, [T])
aclCreateTensorList = _sig("aclCreateTensorList", T, [POINTER(T), c_uint64])
aclDestroyTensorList = _sig("aclDestroyTensorList", c_int, [T])
aclCreateIntArray = _sig("aclCreateIntArray", T, [POINTER(c_int64), c_uint64])
aclDestroyIntArray = _sig("aclDestroyIntArray", c_int, [T])
aclSetAclOpExecutorRepeatable = _sig("aclSetAclOpExecutorRepeatable", c_int, [T])
aclDestroyAclOpExecutor = _sig("aclDestroyAclOpExecutor", c_int, [T])
aclGetRecentErrMsg = _sig("aclGetRecentErrMsg", c_char_p, [])
spfaSetTopK = _sig("spfaSetTopK", c_int, [c_int64])
spfaSetSinkRecent = _sig("spfaSetSinkRecent", c_int, [c_int])
spfaSetPageCap = _sig("spfaSetPageCap", c_int, [c_int])
spfaSetMaxSplit = _sig("spfaSetMaxSplit", c_int, [c_int64])
spfaSetTailSharing = _sig("spfaSetTailSharing", c_int, [c_int])
spfaSetSharedScoring = _sig("spfaSetSharedScoring", c_int, [c_int])
spfaSetWorkgroupGeometry = _sig("spfaSetWorkgroupGeometry", c_int, [c_int])
spfaSetPageSize = _sig("spfaSetPageSize", c_int, [c_int64])
spfaSetActualSeqLengthsDevice = _sig("spfaSetActualSeqLengthsDevice", c_int, [c_void_p, c_int64])
spfaSetTokenPositions = _sig("spfaSetTokenPositions", c_int, [c_int])
spfaSetAntiquantMode = _sig("spfaSetAntiquantMode", c_int, [c_int64])
spfaSetSoftmaxLseOut = _sig("spfaSetSoftmaxLseOut", c_int, [c_void_p, c_int64])
spfaSetSharedPrefix = _sig("spfaSetSharedPrefix", c_int, [c_void_p, c_void_p, c_int64, c_int64])
spfaSetWorkspacePersistent = _sig("spfaSetWorkspacePersistent", c_int, [T, c_int])
spfaSetBatchedSelection = _sig("spfaSetBatchedSelection", c_int, [c_int])
spfaSetKThroughLds = _sig("spfaSetKThroughLds", c_int, [c_int])
spfaSetLibraryCounters = _sig("spfaSetLibraryCounters", c_int, [c_int])
spfaDescribeExecutor = _sig("spfaDescribeExecutor", c_int, [T, ctypes.c_char_p, c_uint64])
spfaKernelLaunches = _sig("spfaKernelLaunches", c_uint64, [])
spfaVersion = _sig("spfaVersion", c_char_p, [])

_WS = [POINTER(c_uint64), POINTER(T)]          # uint64_t* workspaceSize, aclOpExecutor** executor
_EXEC = [c_void_p, c_uint64, T, c_void_p]      # workspace, size, executor, stream

aclnnComputeCentGetWorkspaceSize = _sig("aclnnComputeCentGetWorkspaceSize", c_int, [T] * 3 + _WS)
aclnnComputeCent = _sig("aclnnComputeCent", c_int, _EXEC)
aclnnSelectPositionGetWorkspaceSize = _sig("aclnnSelectPositionGetWorkspaceSize", c_int, [T] * 7 + _WS)
aclnnSelectPosition = _sig("aclnnSelectPosition", c_int, _EXEC)
aclnnCentSelectGetWorkspaceSize = _sig("aclnnCentSelectGetWorkspaceSize", c_int, [T] * 9 + _WS)
aclnnCentSelect = _sig("aclnnCentSelect", c_int, _EXEC)
_ATTRS = [c_int64, c_double, c_char_p, c_int64, c_int64, c_int64]
aclnnSparsePagedAttentionGetWorkspaceSize = _sig(
    "aclnnSparsePagedAttentionGetWorkspaceSize", c_int, [T] * 16 + _ATTRS + [T] + _WS)
aclnnSparsePagedAttention = _sig("aclnnSparsePagedAttention", c_int, _EXEC)
aclnnSparsePagedFusionAttentionGetWorkspaceSize = _sig(
    "aclnnSparsePagedFusionAttentionGetWorkspaceSize", c_int, [T] * 18 + _ATTRS + [T] * 4 + _WS)
aclnnSparsePagedFusionAttention = _sig("aclnnSparsePagedFusionAttention", c_int, _EXEC)
aclnnAssignBlockIdsGetWorkspaceSize = _sig("aclnnAssignBlockIdsGetWorkspaceSize", c_int,
                                           [T] * 7 + [c_int64, c_int64, T] + _WS)
aclnnAssignBlockIds = _sig("aclnnAssignBlockIds", c_int, _EXEC)
aclnnSparsePagedFusionAttentionNoAlloc = _sig("aclnnSparsePagedFusionAttentionNoAlloc", c_int, [T, c_void_p])


def describe(executor) -> dict:
    """The launch plan of an attention executor as a dict (spfaDescribeExecutor)."""
    buf = ctypes.create_string_buffer(512)
    if spfaDescribeExecutor(executor, buf, 512) < 0:
        raise AclError("spfaDescribeExecutor: not an attention executor")
    out = {}
    for part in buf.value.decode().split():
        key, _, val = part.partition("=")
        out[key] = int(val) if val.lstrip("-").isdigit() else val
    return out


class AclError(RuntimeError):
    pass


def check(status: int, what: str):
    if status != 0:
        msg = aclGetRecentErrMsg()
        raise AclError(f"{what} failed: {status}: {msg.decode() if msg else ''}")


def to_acl(t):
    """torch.Tensor -> aclTensor*, or None for an absent operand (None or numel()==0,
    add_custom.cpp:32-35)."""
    if t is None or not isinstance(t, torch.Tensor) or t.numel() == 0:
        return None
    if t.dtype not in _ACL_DTYPE:
        raise AclError(f"Unsupported tensor dtype {t.dtype}")
    n = t.dim()
    dims = (c_int64 * max(n, 1))(*t.shape)
    strides = (c_int64 * max(n, 1))(*t.stride())
    return aclCreateTensor(dims, n, _ACL_DTYPE[t.dtype], strides, 0, ACL_FORMAT_ND, dims, n, c_void_p(t.data_ptr()))


def to_acl_list(tensors):
    descs = [to_acl(t) for t in tensors]
    descs = [d for d in descs if d is not None]
    if not descs:
        return None
    arr = (T * len(descs))(*descs)
    return aclCreateTensorList(arr, len(descs))


def to_int_array(values):
    """list / tuple / int / CPU tensor -> aclIntArray* (host copy)."""
    if values is None:
        return None
    if isinstance(values, torch.Tensor):
        if values.numel() == 0:
            return None
        values = values.detach().to("cpu", torch.int64).reshape(-1).tolist()
    elif isinstance(values, (int,)):
        values = [int(values)]
    else:
        values = [int(v) for v in values]
    if not values:
        return None
    arr = (c_int64 * len(values))(*values)
    return aclCreateIntArray(arr, len(values))


def current_stream(device) -> c_void_p:
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def run(get_ws, exe, args, device, what, keep=()):
    """Phase 1 -> caller-owned workspace -> phase 2 on torch's current stream
    (EXEC_NPU_CMD, pytorch_npu_helper.hpp:721-799)."""
    ws_size = c_uint64(0)
    executor = T()
    ws = None
    ws_ptr = c_void_p(0)
    # both phases run with the tensors' device current: phase 1 plans for the current device (CU count, residency,
    # LDS limit of the kernel) and phase 2 refuses to launch on another one
    with torch.cuda.device(device):
        check(get_ws(*args, ctypes.byref(ws_size), ctypes.byref(executor)), what + "GetWorkspaceSize")
        if ws_size.value > 0:
            ws = torch.empty(ws_size.value, dtype=torch.uint8, device=device)
            ws_ptr = c_void_p(ws.data_ptr())
        status = exe(ws_ptr, ws_size.value, executor, current_stream(device))
    if ws is not None:
        ws.record_stream(torch.cuda.current_stream(device))
    check(status, what)
    return ws


def require_gpu(*tensors):
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.numel() > 0 and not t.is_cuda:
            raise AclError("custom_ops operands must live on the GPU (there is no CPU implementation)")
