"""`custom_ops.*` operator surface on MI355X.

Same function names, positional argument order and return tuples as the reference's
extension/custom_ops/add_custom.py (:41-135); every call goes through the aclnn-style C-ABI of
libcust_opapi.so on torch's current HIP stream.  Optional operands follow the reference's
convention: an empty tensor (numel() == 0) means "absent" (extension/csrc/add_custom.cpp:32-35).

Keyword-only extras (not in the reference, which hard-codes them; they travel through one-shot per-thread settings of
the C-ABI that the very next phase 1 consumes):
  k             top-k clusters (reference: 64, add_custom.cpp:940 / cent_select_tiling.cpp:58)
  sink_recent   also select logical pages 0, pageLen-2, pageLen-1 (cent_select.cpp:350-362)
  max_pages_cap clip page_position_length to pageLen // 8 (cent_select.cpp:218,410-414)
  max_page_num  slots per page_position row (reference: 256, add_custom.cpp:958,987)
  page_size     page size of cent_select / select_position (reference: 128, cent_select.cpp:17)
The two attention ops also take the operands of the reference KERNEL that its aclnn signatures for the sparse ops leave
out (sparse_paged_fusion_attention_split_Bbn2s2_Us2.h:1418-1439, 3130-3159):
  softmax_lse               float32 tensor of B * num_heads values, filled with every row's softmax log-sum-exp
  key_shared_prefix, value_shared_prefix, actual_shared_prefix_len
                            K / V rows of a prefix that every row attends in front of its pages: [1, S, Hkv * D] for a
                            [blocks, block_size, Hkv * D] cache, [1, Hkv, S, D] for a [blocks, Hkv, block_size, D] cache
  antiquant_mode            0: antiquant_scale / antiquant_offset per channel [2, Hkv, D] or per tensor [2] (query dtype);
                            1: per token, float32 [2, B, S] over the columns of the gathered sequence;
                            2: per tensor and kv head, [2, Hkv] (query dtype); 3: per token and kv head, float32
                            [2, B, Hkv, S] (the reference's modes 4 and 5 keep the parameters in pages: not supported)
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple, Union

import torch

from . import _abi

_SENTINEL = 0x7FFFFFFF


def _destroy(tensors=(), lists=(), arrays=()):
    for d in tensors:
        if d is not None:
            _abi.aclDestroyTensor(d)
    for d in lists:
        if d is not None:
            _abi.aclDestroyTensorList(d)
    for d in arrays:
        if d is not None:
            _abi.aclDestroyIntArray(d)


def _set_knobs(k=None, sink_recent=False, max_pages_cap=False, page_size=128):
    if k is not None:
        _abi.check(_abi.spfaSetTopK(int(k)), "spfaSetTopK")
    _abi.check(_abi.spfaSetSinkRecent(int(bool(sink_recent))), "spfaSetSinkRecent")
    _abi.check(_abi.spfaSetPageCap(int(bool(max_pages_cap))), "spfaSetPageCap")
    _abi.check(_abi.spfaSetPageSize(int(page_size)), "spfaSetPageSize")


def _clear_knobs():
    """Back to the defaults: a call that failed before its phase 1 must not leave one-shot settings armed."""
    _abi.spfaSetTopK(64)
    _abi.spfaSetSinkRecent(0)
    _abi.spfaSetPageCap(0)
    _abi.spfaSetPageSize(128)


def incre_flash_attention_v4(*args, **kwargs):
    """Stock dense IncreFlashAttention (add_custom.py:10-40): not part of the sparse paged path."""
    raise NotImplementedError("incre_flash_attention_v4 is outside the sparse paged attention path")


def compute_cent(query: torch.Tensor, l1_cent: torch.Tensor, *, k: int = 64) -> torch.Tensor:
    """indices[b,h,:k] of the k best centroids of kv head h//G for q[b,h] (add_custom.py:77-81;
    binding add_custom.cpp:937-953 allocates [B,Hq,64] int32)."""
    _abi.require_gpu(query, l1_cent)
    out = torch.empty((query.shape[0], query.shape[1], int(k)), dtype=torch.int32, device=query.device)
    descs = [_abi.to_acl(query), _abi.to_acl(l1_cent), _abi.to_acl(out)]
    try:
        _abi.run(_abi.aclnnComputeCentGetWorkspaceSize, _abi.aclnnComputeCent, descs, query.device, "aclnnComputeCent")
    finally:
        _destroy(descs)
    return out


def select_position(block_ids: torch.Tensor, block_table: torch.Tensor, seq_len: torch.Tensor,
                    indices: torch.Tensor, *, max_page_num: int = 256, page_size: int = 128
                    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(page_position [B,Hq,256], page_position_length [B,Hq,8], block_table_gather [B,Hq,maxPage])
    (add_custom.py:83-87; binding add_custom.cpp:955-977)."""
    _abi.require_gpu(block_ids, block_table, seq_len, indices)
    B, Hq = indices.shape[0], indices.shape[1]
    dev = block_ids.device
    pos = torch.empty((B, Hq, int(max_page_num)), dtype=torch.int32, device=dev)
    length = torch.empty((B, Hq, 8), dtype=torch.int32, device=dev)
    gather = torch.empty((B, Hq, block_table.shape[1]), dtype=torch.int32, device=dev)
    descs = [_abi.to_acl(t) for t in (block_ids, block_table, seq_len, indices, pos, length, gather)]
    try:
        _set_knobs(page_size=page_size)   # one-shot settings: right in front of the phase 1 that consumes them
        _abi.run(_abi.aclnnSelectPositionGetWorkspaceSize, _abi.aclnnSelectPosition, descs, dev, "aclnnSelectPosition")
    finally:
        _clear_knobs()
        _destroy(descs)
    return pos, length, gather


def cent_select(query: torch.Tensor, l1_cent: torch.Tensor, block_ids: torch.Tensor, block_table: torch.Tensor,
                seq_len: torch.Tensor, *, k: int = 64, sink_recent: bool = False, max_pages_cap: bool = False,
                max_page_num: int = 256, page_size: int = 128
                ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(page_position [B,Hq,256] i32, page_position_length [B,Hq,8] i32, max_page_position_length [B,8] i64)
    with B = seq_len.shape[0] (add_custom.py:89-93; binding add_custom.cpp:979-1008)."""
    _abi.require_gpu(query, l1_cent, block_ids, block_table, seq_len)
    B, Hq = seq_len.shape[0], query.shape[1]
    dev = query.device
    pos = torch.empty((B, Hq, int(max_page_num)), dtype=torch.int32, device=dev)
    length = torch.empty((B, Hq, 8), dtype=torch.int32, device=dev)
    max_len = torch.empty((B, 8), dtype=torch.int64, device=dev)
    descs = [_abi.to_acl(t) for t in (query, l1_cent, block_ids, block_table, seq_len)]
    descs += [None]  # importance: allocated but never read by the reference (add_custom.cpp:995)
    descs += [_abi.to_acl(t) for t in (pos, length, max_len)]
    try:
        _set_knobs(k, sink_recent, max_pages_cap, page_size)
        _abi.run(_abi.aclnnCentSelectGetWorkspaceSize, _abi.aclnnCentSelect, descs, dev, "aclnnCentSelect")
    finally:
        _clear_knobs()
        _destroy(descs)
    return pos, length, max_len


def _seq_lengths_arg(actual_seq_lengths, device):
    """Host list / CPU tensor -> aclIntArray; a GPU tensor is handed over as a device pointer so the
    call does not synchronise (the reference copies it to the host, add_custom.cpp:113)."""
    if isinstance(actual_seq_lengths, torch.Tensor) and actual_seq_lengths.is_cuda and actual_seq_lengths.numel() > 0:
        dev_lengths = actual_seq_lengths.to(torch.int64).contiguous()
        _abi.check(_abi.spfaSetActualSeqLengthsDevice(dev_lengths.data_ptr(), dev_lengths.numel()),
                   "spfaSetActualSeqLengthsDevice")
        return None, dev_lengths
    return _abi.to_int_array(actual_seq_lengths), None


def _set_attention_extras(softmax_lse, key_shared_prefix, value_shared_prefix, actual_shared_prefix_len, key,
                          num_key_value_heads, block_size, antiquant_mode=0):
    """Arm the one-shot operands that the aclnn signatures have no slot for; call right in front of phase 1."""
    _abi.check(_abi.spfaSetAntiquantMode(int(antiquant_mode)), "spfaSetAntiquantMode")
    if softmax_lse is not None:
        if softmax_lse.dtype != torch.float32 or not softmax_lse.is_contiguous() or not softmax_lse.is_cuda:
            raise _abi.AclError("softmax_lse must be a contiguous float32 GPU tensor")
        _abi.check(_abi.spfaSetSoftmaxLseOut(softmax_lse.data_ptr(), softmax_lse.numel()), "spfaSetSoftmaxLseOut")
    if key_shared_prefix is not None or value_shared_prefix is not None:
        kp, vp = key_shared_prefix, value_shared_prefix
        if kp is None or vp is None or kp.shape != vp.shape or kp.dtype != key.dtype or vp.dtype != key.dtype:
            raise _abi.AclError("key / value shared prefix must come together, equal in shape, in the KV cache's dtype")
        if not (kp.is_cuda and vp.is_cuda and kp.is_contiguous() and vp.is_contiguous()) or kp.shape[0] != 1:
            raise _abi.AclError("shared prefix tensors must be contiguous GPU tensors with batch 1")
        # the cache is [blocks, bs, Hkv * D], [blocks, bs, Hkv, D] (both token-major) or [blocks, Hkv, bs, D]; the C side
        # tells the last two apart by dims[1] == block_size first, so do the same here
        bnsd = key.dim() == 4 and not (key.shape[1] == block_size and key.shape[2] == num_key_value_heads)
        tokens = kp.shape[2] if (bnsd and kp.dim() == 4) else kp.shape[1]
        row = key.shape[-1] * (num_key_value_heads if key.dim() == 4 else 1)   # elements of one token over all kv heads
        if kp.numel() != tokens * row:
            raise _abi.AclError("shared prefix must be [1, S, Hkv * D] / [1, S, Hkv, D] (token-major cache) or "
                                "[1, Hkv, S, D] ([blocks, Hkv, block_size, D] cache)")
        n = tokens if actual_shared_prefix_len is None else int(actual_shared_prefix_len)
        _abi.check(_abi.spfaSetSharedPrefix(kp.data_ptr(), vp.data_ptr(), int(tokens), n), "spfaSetSharedPrefix")


def _clear_attention_extras():
    _abi.spfaSetAntiquantMode(0)
    _abi.spfaSetSoftmaxLseOut(None, 0)
    _abi.spfaSetSharedPrefix(None, None, 0, 0)


def _default_out_dtype(query, quant_scale2):
    """The output has the query's dtype unless an output quantisation scale asks for int8
    (sparse_paged_fusion_attention_tiling.cc:569-605)."""
    if isinstance(quant_scale2, torch.Tensor) and quant_scale2.numel() > 0:
        return torch.int8
    return query.dtype


def sparse_paged_attention(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, pse_shift: torch.Tensor,
                           attention_mask: torch.Tensor, actual_seq_lengths, dequant_scale1: torch.Tensor,
                           quant_scale1: torch.Tensor, dequant_scale2: torch.Tensor, quant_scale2: torch.Tensor,
                           quant_offset2: torch.Tensor, antiquant_scale: torch.Tensor,
                           antiquant_offset: torch.Tensor, blocktable: torch.Tensor, kv_padding_size: torch.Tensor,
                           blockposition: torch.Tensor, num_heads: int, scale_value: float, input_layout: str,
                           num_key_value_heads: int, block_size: int, inner_precise: int,
                           *, out_dtype: Optional[torch.dtype] = None, token_level: bool = False,
                           softmax_lse: Optional[torch.Tensor] = None, key_shared_prefix: Optional[torch.Tensor] = None,
                           value_shared_prefix: Optional[torch.Tensor] = None,
                           actual_shared_prefix_len: Optional[int] = None, antiquant_mode: int = 0) -> torch.Tensor:
    """Decode attention over the pages listed in blockposition (add_custom.py:41-75; binding
    sparse_paged_attention_impl_npu wraps key/value into one-tensor lists and returns a tensor shaped
    like query).  out_dtype=torch.float32 (extra) returns the un-rounded result.

    pse_shift [B | 1, num_heads, 1, S] (query dtype) is added to the scaled scores, attention_mask [B, ..., S]
    (bool / int8 / uint8, non zero = masked) replaces them by -1e12; both run over the columns of the gathered
    sequence, list slot * block_size + token.  quant_scale2 / quant_offset2 (float32 or bfloat16; one value or
    num_heads * D) turn the output into int8.  antiquant_scale / antiquant_offset may be [2, Hkv, D] (per channel)
    or [2] (per tensor).  dequant_scale1, quant_scale1 and dequant_scale2 must be empty; kv_padding_size is ignored (paged KV
    has no left padding, sparse_paged_fusion_attention_tiling_check.cc:590-594).

    token_level=True (extra; the variant sketched in the reference's tests/ifa_v5_case_tkp.py:53-54): blockposition
    [B, num_heads | num_key_value_heads, max_tokens] holds logical TOKEN indices (sentinel terminated; token t is row
    t % block_size of page blocktable[b][t // block_size]) and actual_seq_lengths counts tokens."""
    _abi.require_gpu(query, key, value, blocktable, blockposition)
    dev = query.device
    out = torch.empty(query.shape, dtype=out_dtype or _default_out_dtype(query, quant_scale2), device=dev)
    tens, opt, quant, lists, arrays, dev_lengths = [], [], [], [], [], None
    try:
        tens = [_abi.to_acl(query)]
        lists = [_abi.to_acl_list([key]), _abi.to_acl_list([value])]
        opt = [_abi.to_acl(t) for t in (pse_shift, attention_mask)]
        quant = [_abi.to_acl(t) for t in (dequant_scale1, quant_scale1, dequant_scale2, quant_scale2, quant_offset2,
                                          antiquant_scale, antiquant_offset, blocktable, kv_padding_size,
                                          blockposition)]
        tens.append(_abi.to_acl(out))
        # last, right in front of phase 1 (which consumes it): the device-resident lengths are a one-shot setting
        seq_arr, dev_lengths = _seq_lengths_arg(actual_seq_lengths, dev)
        arrays = [seq_arr]
        _abi.check(_abi.spfaSetTokenPositions(1 if token_level else 0), "spfaSetTokenPositions")
        _set_attention_extras(softmax_lse, key_shared_prefix, value_shared_prefix, actual_shared_prefix_len, key,
                              int(num_key_value_heads), int(block_size), antiquant_mode)
        args = tens[:1] + lists + opt + [seq_arr] + quant + [
            int(num_heads), float(scale_value), str(input_layout).encode(), int(num_key_value_heads), int(block_size),
            int(inner_precise), tens[1]]
        _abi.run(_abi.aclnnSparsePagedAttentionGetWorkspaceSize, _abi.aclnnSparsePagedAttention, args, dev,
                 "aclnnSparsePagedAttention")
    finally:
        _abi.spfaSetActualSeqLengthsDevice(None, 0)
        _abi.spfaSetTokenPositions(0)
        _clear_attention_extras()
        _destroy(tens + opt + quant, lists, arrays)
    if dev_lengths is not None:
        dev_lengths.record_stream(torch.cuda.current_stream(dev))
    return out


def sparse_paged_fusion_attention(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                                  pse_shift: torch.Tensor, attention_mask: torch.Tensor, actual_seq_lengths,
                                  dequant_scale1: torch.Tensor, quant_scale1: torch.Tensor,
                                  dequant_scale2: torch.Tensor, quant_scale2: torch.Tensor,
                                  quant_offset2: torch.Tensor, antiquant_scale: torch.Tensor,
                                  antiquant_offset: torch.Tensor, blocktable: torch.Tensor,
                                  kv_padding_size: torch.Tensor, l1_cent: torch.Tensor, block_ids: torch.Tensor,
                                  total_seq_len: torch.Tensor, block_position: torch.Tensor,
                                  page_position_length: torch.Tensor, max_page_position_length: torch.Tensor,
                                  num_heads: int, scale_value: float, input_layout: str, num_key_value_heads: int,
                                  block_size: int, inner_precise: int, *, k: int = 64, sink_recent: bool = False,
                                  max_pages_cap: bool = False, out_dtype: Optional[torch.dtype] = None,
                                  softmax_lse: Optional[torch.Tensor] = None,
                                  key_shared_prefix: Optional[torch.Tensor] = None,
                                  value_shared_prefix: Optional[torch.Tensor] = None,
                                  actual_shared_prefix_len: Optional[int] = None, antiquant_mode: int = 0
                                  ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """CentSelect + sparse paged attention in one kernel launch (add_custom.py:96-135).  Returns
    (attention_out, block_position, max_page_position_length) — the 3-tuple of the registered schema
    (extension/csrc/registration.cpp:50-56); block_position, page_position_length and
    max_page_position_length are filled in place.  actual_seq_lengths is only a tiling hint in the
    reference (sparse_paged_fusion_attention.cpp:68) and is ignored."""
    _abi.require_gpu(query, key, value, blocktable, l1_cent, block_ids, total_seq_len)
    dev = query.device
    out = torch.empty(query.shape, dtype=out_dtype or _default_out_dtype(query, quant_scale2), device=dev)
    tens = [_abi.to_acl(query)]
    kl, vl = _abi.to_acl_list([key]), _abi.to_acl_list([value])
    mid = [_abi.to_acl(t) for t in (pse_shift, attention_mask)]
    rest = [_abi.to_acl(t) for t in (dequant_scale1, quant_scale1, dequant_scale2, quant_scale2, quant_offset2,
                                     antiquant_scale, antiquant_offset, blocktable, kv_padding_size, l1_cent,
                                     block_ids, total_seq_len)]
    outs = [_abi.to_acl(t) for t in (block_position, page_position_length, max_page_position_length, out)]
    args = tens + [kl, vl] + mid + [None] + rest + [
        int(num_heads), float(scale_value), str(input_layout).encode(), int(num_key_value_heads), int(block_size),
        int(inner_precise)] + outs
    try:
        # one-shot settings, armed right in front of the phase 1 that consumes them (the checks that can raise first)
        _set_attention_extras(softmax_lse, key_shared_prefix, value_shared_prefix, actual_shared_prefix_len, key,
                              int(num_key_value_heads), int(block_size), antiquant_mode)
        _set_knobs(k, sink_recent, max_pages_cap)
        _abi.run(_abi.aclnnSparsePagedFusionAttentionGetWorkspaceSize, _abi.aclnnSparsePagedFusionAttention, args,
                 dev, "aclnnSparsePagedFusionAttention")
    finally:
        _clear_attention_extras()
        _clear_knobs()
        _destroy(tens + mid + rest + outs, [kl, vl])
    return out, block_position, max_page_position_length


def assign_block_ids(key: torch.Tensor, block_table: torch.Tensor, seq_len: torch.Tensor, l1_cent: torch.Tensor,
                     block_ids: torch.Tensor, first_new_page: torch.Tensor, *, block_size: int = 128,
                     max_new_pages: int = 0, antiquant_scale: Optional[torch.Tensor] = None,
                     antiquant_offset: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Cluster ids for newly filled KV pages, written into block_ids [Hkv, kvPageLen] in place (returned).

    Not in the reference, which only consumes block_ids (README.md:33): for every sequence b and logical page i
    with first_new_page[b] <= i < ceil(seq_len[b] / block_size) (at most max_new_pages per sequence; 0 = all) and every
    kv head, the mean K row of the page (rounded to l1_cent's dtype) is scored against the head's centroids like a
    query row and the page gets the best cluster's id (ties: smallest id).  A page whose block_table entry lies outside
    the KV pool (the usual -1 of an unassigned slot) is skipped.  A decode loop calls this when a page
    fills up and advances first_new_page itself."""
    _abi.require_gpu(key, block_table, seq_len, l1_cent, block_ids, first_new_page)
    dev = key.device
    kl = _abi.to_acl_list([key])
    descs = [_abi.to_acl(t) for t in (block_table, seq_len, first_new_page, l1_cent, antiquant_scale, antiquant_offset)]
    ids_d = _abi.to_acl(block_ids)
    try:
        _abi.run(_abi.aclnnAssignBlockIdsGetWorkspaceSize, _abi.aclnnAssignBlockIds,
                 [kl] + descs + [int(block_size), int(max_new_pages), ids_d], dev, "aclnnAssignBlockIds")
    finally:
        _destroy(descs + [ids_d], [kl])
    return block_ids


class PreparedFusion:
    """Plan once, launch many: keeps the executor of sparse_paged_fusion_attention alive
    (aclSetAclOpExecutorRepeatable, the pattern of extension/csrc/add_custom.cpp:847,886-895) so a
    decode loop pays descriptor conversion and validation once.  Tensors are captured by pointer:
    update their contents in place between launches."""

    def __init__(self, query, key, value, blocktable, l1_cent, block_ids, total_seq_len, block_position,
                 page_position_length, max_page_position_length, num_heads, scale_value, num_key_value_heads,
                 block_size, *, antiquant_scale=None, antiquant_offset=None, k=64, sink_recent=False,
                 max_pages_cap=False, out_dtype=None, input_layout="BNSD", persistent_workspace=True):
        import ctypes
        _abi.require_gpu(query, key, value, blocktable, l1_cent, block_ids, total_seq_len)
        self.device = query.device
        self.out = torch.empty(query.shape, dtype=out_dtype or query.dtype, device=self.device)
        self._keep = (query, key, value, blocktable, l1_cent, block_ids, total_seq_len, block_position,
                      page_position_length, max_page_position_length, antiquant_scale, antiquant_offset)
        tens = [_abi.to_acl(query)]
        kl, vl = _abi.to_acl_list([key]), _abi.to_acl_list([value])
        rest = [None] * 5 + [_abi.to_acl(t) for t in (antiquant_scale, antiquant_offset, blocktable)] + [None] + \
            [_abi.to_acl(t) for t in (l1_cent, block_ids, total_seq_len)]
        outs = [_abi.to_acl(t) for t in (block_position, page_position_length, max_page_position_length, self.out)]
        args = tens + [kl, vl, None, None, None] + rest + [
            int(num_heads), float(scale_value), str(input_layout).encode(), int(num_key_value_heads),
            int(block_size), 1] + outs
        ws_size = ctypes.c_uint64(0)
        self._exec = _abi.T()
        try:
            with torch.cuda.device(self.device):   # the plan is made for, and bound to, the tensors' device
                _set_knobs(k, sink_recent, max_pages_cap)
                _abi.check(_abi.aclnnSparsePagedFusionAttentionGetWorkspaceSize(*args, ctypes.byref(ws_size),
                                                                                ctypes.byref(self._exec)),
                           "aclnnSparsePagedFusionAttentionGetWorkspaceSize")
        finally:
            _clear_knobs()
            _destroy(tens + rest + outs, [kl, vl])
        _abi.check(_abi.aclSetAclOpExecutorRepeatable(self._exec), "aclSetAclOpExecutorRepeatable")
        # zeroed here, once: whatever the first launch is (eager, or recorded into a graph) it finds clean counters
        self._ws = torch.zeros(max(int(ws_size.value), 1), dtype=torch.uint8, device=self.device)
        self._ws_size = int(ws_size.value)
        # the workspace belongs to this plan alone, so its counters stay as every launch leaves them (zeroed): after
        # the first launch each run() is exactly one kernel
        _abi.check(_abi.spfaSetWorkspacePersistent(self._exec, 1 if persistent_workspace else 0),
                   "spfaSetWorkspacePersistent")

    def describe(self) -> dict:
        """The launch plan (kernel form, geometry, how the work is dealt): see spfaDescribeExecutor."""
        return _abi.describe(self._exec)

    def run(self) -> torch.Tensor:
        """Enqueue one fused launch on torch's current stream; returns the (reused) output tensor."""
        with torch.cuda.device(self.device):
            _abi.check(_abi.aclnnSparsePagedFusionAttention(self._ws.data_ptr() if self._ws_size else None,
                                                            self._ws_size, self._exec,
                                                            _abi.current_stream(self.device)),
                       "aclnnSparsePagedFusionAttention")
        return self.out

    def close(self):
        if self._exec:
            _abi.aclDestroyAclOpExecutor(self._exec)
            self._exec = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
