"""MI355X-native `custom_ops` package: the reference's operator surface
(extension/custom_ops/__init__.py:1 re-exports add_custom) over hand-written gfx950 kernels."""
from .add_custom import (PreparedFusion, assign_block_ids, cent_select, compute_cent, incre_flash_attention_v4, select_position,  # noqa: F401
                         sparse_paged_attention, sparse_paged_fusion_attention)

__all__ = ["PreparedFusion", "assign_block_ids", "cent_select", "compute_cent", "incre_flash_attention_v4", "select_position",
           "sparse_paged_attention", "sparse_paged_fusion_attention", "set_dispatch_defaults"]

# torch.ops.myops.* (the reference registers its schemas when its extension is imported,
# extension/csrc/registration.cpp:43-57)
from .registration import register as _register, set_dispatch_defaults  # noqa: E402,F401

_register()
