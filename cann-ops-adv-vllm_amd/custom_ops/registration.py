"""`torch.ops.myops.*` — the dispatcher-level surface of the reference (extension/csrc/registration.cpp:43-57,
implementations bound to PrivateUse1 at :60-68) bound to the HIP device key instead.

The schemas are the reference's (same operator names, argument names and order, `Tensor[]` key/value lists,
`Tensor actual_seq_lengths`, three-tensor returns); `incre_flash_attention_v4` is not part of the sparse paged
path and is not registered.  Each implementation forwards to the function of the same name in
`custom_ops.add_custom`, i.e. to the aclnn C-ABI; the extras that the reference hard-codes keep its values
(k = 64, 256 page slots, 128-token pages for the two ops without a block_size attribute).
"""
from __future__ import annotations

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
        num_key_value_heads, block_size, inner_precise)


_IMPLS = {
    "sparse_paged_attention": _sparse_paged_attention,
    "compute_cent": lambda query, l1_cent: _ops.compute_cent(query, l1_cent),
    "select_position": lambda block_ids, block_table, seq_len, indices:
        _ops.select_position(block_ids, block_table, seq_len, indices),
    "cent_select": lambda query, l1_cent, block_ids, block_table, seq_len:
        _ops.cent_select(query, l1_cent, block_ids, block_table, seq_len),
    "sparse_paged_fusion_attention": _sparse_paged_fusion_attention,
}

_def_lib = None
_impl_lib = None


def register() -> None:
    """Define the `myops` schemas and bind them for GPU tensors.  Idempotent; raises if another extension
    already owns the `myops` namespace with different schemas."""
    global _def_lib, _impl_lib
    if _def_lib is not None:
        return
    _def_lib = torch.library.Library("myops", "DEF")
    for schema in _SCHEMAS:
        _def_lib.define(schema)
    _impl_lib = torch.library.Library("myops", "IMPL", "CUDA")   # the HIP device key of PyTorch-ROCm
    for name, fn in _IMPLS.items():
        _impl_lib.impl(name, fn)
