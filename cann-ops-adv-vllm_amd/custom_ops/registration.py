"""`torch.ops.myops.*` — the dispatcher-level surface of the reference (extension/csrc/registration.cpp:43-57,
implementations bound to PrivateUse1 at :60-68) bound to the HIP device key instead.

The schemas are the reference's (same operator names, argument names and order, `Tensor[]` key/value lists,
`Tensor actual_seq_lengths`, three-tensor returns); `incre_flash_attention_v4` is not part of the sparse paged
path and is not registered.  Each implementation forwards to the function of the same name in
`custom_ops.add_custom`, i.e. to the aclnn C-ABI; the extras that the reference hard-codes keep its values
(k = 64, 256 page slots, 128-token pages for the two ops without a block_size attribute).
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch

from . import add_custom as _ops

_SCHEMAS = (
    "sparse_paged_attention(Tensor query, Tensor[] key_list, Tensor[] value_list, Tensor pse_shift, "
    "Tensor attention_mask, Tensor actual_seq_lengths, Tensor dequant_scale1, Tensor quant_scale1, "
    "Tensor dequant_scale2, Tensor quant_scale2, Tensor quant_offset2, Tensor antiquant_scale, "
    "Tensor antiquant_offset, Tensor blocktable, Tensor kv_padding_size, Tensor blockposition, int num_heads, "
    "float scale_value, str input_layout, int num_key_value_heads, int block_size, int inner_precise) -> Tensor",
    "compute_cent(Tensor query, Tensor l1_cent) -> Tensor",
    "select_position(Tensor block_ids, Tensor block_table, Tensor seq_len, Tensor indices) -> (Tensor, Tensor, Tensor)",
    "cent_select(Tensor query, Tensor l1_cent, Tensor block_ids, Tensor block_table, Tensor seq_len) "
    "-> (Tensor, Tensor, Tensor)",
    "sparse_paged_fusion_attention(Tensor query, Tensor[] key_list, Tensor[] value_list, Tensor pse_shift, "
    "Tensor attention_mask, Tensor actual_seq_lengths, Tensor dequant_scale1, Tensor quant_scale1, "
    "Tensor dequant_scale2, Tensor quant_scale2, Tensor quant_offset2, Tensor antiquant_scale, "
    "Tensor antiquant_offset, Tensor blocktable, Tensor kv_padding_size, Tensor l1_cent, Tensor block_ids, "
    "Tensor total_seq_len, Tensor(a!) block_position, Tensor(b!) page_position_length, "
    "Tensor(c!) max_page_position_length, int num_heads, float scale_value, str input_layout, "
    "int num_key_value_heads, int block_size, int inner_precise) -> (Tensor, Tensor(a!), Tensor(c!))",
)


# What the dispatcher-level ops use for the values the reference hard-codes.  The defaults reproduce the reference's
# Python/CPU path (no sink/recent pages, no pageLen/8 cap); SPFA_KERNEL_FAITHFUL=1 in the environment, or
# set_dispatch_defaults(sink_recent=True, max_pages_cap=True), selects what its device kernel does instead
# (cent_select.cpp:350-362,410-414), so torch.ops.myops callers can reach either behaviour.
_DEFAULTS = dict(k=64, sink_recent=False, max_pages_cap=False, max_page_num=256, page_size=128)
if os.environ.get("SPFA_KERNEL_FAITHFUL", "0") not in ("", "0"):
    _DEFAULTS.update(sink_recent=True, max_pages_cap=True)


def set_dispatch_defaults(**kwargs) -> dict:
    """Change the extras that `torch.ops.myops.*` pass on (k, sink_recent, max_pages_cap, max_page_num, page_size);
    returns the previous values."""
    unknown = set(kwargs) - set(_DEFAULTS)
    if unknown:
        raise TypeError(f"unknown dispatch default(s): {sorted(unknown)}")
    before = dict(_DEFAULTS)
    _DEFAULTS.update(kwargs)
    return before


def _one(tensors: List[torch.Tensor], what: str) -> torch.Tensor:
    # the reference passes one-tensor lists (add_custom.py wraps key/value, add_custom.cpp:79-102)
    if len(tensors) != 1:
        raise RuntimeError(f"{what}: expected a list with one cache tensor, got {len(tensors)}")
    return tensors[0]


def _sparse_paged_attention(query, key_list, value_list, pse_shift, attention_mask, actual_seq_lengths,
                            dequant_scale1, quant_scale1, dequant_scale2, quant_scale2, quant_offset2,
                            antiquant_scale, antiquant_offset, blocktable, kv_padding_size, blockposition,
                            num_heads, scale_value, input_layout, num_key_value_heads, block_size, inner_precise):
    return _ops.sparse_paged_attention(
        query, _one(key_list, "key_list"), _one(value_list, "value_list"), pse_shift, attention_mask,
        actual_seq_lengths, dequant_scale1, quant_scale1, dequant_scale2, quant_scale2, quant_offset2,
        antiquant_scale, antiquant_offset, blocktable, kv_padding_size, blockposition, num_heads, scale_value,
        input_layout, num_key_value_heads, block_size, inner_precise)


def _sparse_paged_fusion_attention(query, key_list, value_list, pse_shift, attention_mask, actual_seq_lengths,
                                   dequant_scale1, quant_scale1, dequant_scale2, quant_scale2, quant_offset2,
                                   antiquant_scale, antiquant_offset, blocktable, kv_padding_size, l1_cent, block_ids,
                                   total_seq_len, block_position, page_position_length, max_page_position_length,
                                   num_heads, scale_value, input_layout, num_key_value_heads, block_size,
                                   inner_precise) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    return _ops.sparse_paged_fusion_attention(
        query, _one(key_list, "key_list"), _one(value_list, "value_list"), pse_shift, attention_mask,
        actual_seq_lengths, dequant_scale1, quant_scale1, dequant_scale2, quant_scale2, quant_offset2,
        antiquant_scale, antiquant_offset, blocktable, kv_padding_size, l1_cent, block_ids, total_seq_len,
        block_position, page_position_length, max_page_position_length, num_heads, scale_value, input_layout,
        num_key_value_heads, block_size, inner_precise, k=_DEFAULTS["k"], sink_recent=_DEFAULTS["sink_recent"],
        max_pages_cap=_DEFAULTS["max_pages_cap"])


_IMPLS = {
    "sparse_paged_attention": _sparse_paged_attention,
    "compute_cent": lambda query, l1_cent: _ops.compute_cent(query, l1_cent, k=_DEFAULTS["k"]),
    "select_position": lambda block_ids, block_table, seq_len, indices:
        _ops.select_position(block_ids, block_table, seq_len, indices, max_page_num=_DEFAULTS["max_page_num"],
                             page_size=_DEFAULTS["page_size"]),
    "cent_select": lambda query, l1_cent, block_ids, block_table, seq_len:
        _ops.cent_select(query, l1_cent, block_ids, block_table, seq_len, k=_DEFAULTS["k"],
                         sink_recent=_DEFAULTS["sink_recent"], max_pages_cap=_DEFAULTS["max_pages_cap"],
                         max_page_num=_DEFAULTS["max_page_num"], page_size=_DEFAULTS["page_size"]),
    "sparse_paged_fusion_attention": _sparse_paged_fusion_attention,
}


# Shape-only ("meta") implementations, the counterpart of the torch_npu fork's
# op_plugin/python/meta/_meta_registrations.py:38-93: they let the ops be traced (torch.compile, FakeTensor, meta
# tensors) without running a kernel.  Output shapes and dtypes are those of the real implementations above.
def _attn_out_like(query, quant_scale2):
    return torch.empty_like(query, dtype=torch.int8 if quant_scale2.numel() > 0 else query.dtype)


def _meta_sparse_paged_attention(query, key_list, value_list, pse_shift, attention_mask, actual_seq_lengths,
                                 dequant_scale1, quant_scale1, dequant_scale2, quant_scale2, quant_offset2,
                                 antiquant_scale, antiquant_offset, blocktable, kv_padding_size, blockposition,
                                 num_heads, scale_value, input_layout, num_key_value_heads, block_size, inner_precise):
    return _attn_out_like(query, quant_scale2)


def _meta_fusion(query, key_list, value_list, pse_shift, attention_mask, actual_seq_lengths, dequant_scale1,
                 quant_scale1, dequant_scale2, quant_scale2, quant_offset2, antiquant_scale, antiquant_offset,
                 blocktable, kv_padding_size, l1_cent, block_ids, total_seq_len, block_position,
                 page_position_length, max_page_position_length, num_heads, scale_value, input_layout,
                 num_key_value_heads, block_size, inner_precise):
    return _attn_out_like(query, quant_scale2), block_position, max_page_position_length


def _meta_select_position(block_ids, block_table, seq_len, indices):
    B, Hq = indices.shape[0], indices.shape[1]
    new = lambda *shape: torch.empty(shape, dtype=torch.int32, device=indices.device)
    return new(B, Hq, _DEFAULTS["max_page_num"]), new(B, Hq, 8), new(B, Hq, block_table.shape[1])


def _meta_cent_select(query, l1_cent, block_ids, block_table, seq_len):
    B, Hq = seq_len.shape[0], query.shape[1]   # batch from seq_len (cent_select_tiling.cpp:32,53)
    return (torch.empty((B, Hq, _DEFAULTS["max_page_num"]), dtype=torch.int32, device=query.device),
            torch.empty((B, Hq, 8), dtype=torch.int32, device=query.device),
            torch.empty((B, 8), dtype=torch.int64, device=query.device))


_META = {
    "sparse_paged_attention": _meta_sparse_paged_attention,
    "compute_cent": lambda query, l1_cent: torch.empty((query.shape[0], query.shape[1], _DEFAULTS["k"]),
                                                       dtype=torch.int32, device=query.device),
    "select_position": _meta_select_position,
    "cent_select": _meta_cent_select,
    "sparse_paged_fusion_attention": _meta_fusion,
}

_def_lib = None
_impl_lib = None
_meta_lib = None


def register() -> None:
    """Define the `myops` schemas and bind them for GPU tensors.  Idempotent; raises if another extension
    already owns the `myops` namespace with different schemas."""
    global _def_lib, _impl_lib, _meta_lib
    if _def_lib is not None:
        return
    _def_lib = torch.library.Library("myops", "DEF")
    for schema in _SCHEMAS:
        _def_lib.define(schema)
    _impl_lib = torch.library.Library("myops", "IMPL", "CUDA")   # the HIP device key of PyTorch-ROCm
    for name, fn in _IMPLS.items():
        _impl_lib.impl(name, fn)
    _meta_lib = torch.library.Library("myops", "IMPL", "Meta")
    for name, fn in _META.items():
        _meta_lib.impl(name, fn)
