/*
 * CentSelect = ComputeCent + SelectPosition fused, plus max_page_position_length[b,0..7] =
 * max_h(page_position_length[b,h]) * 128.
 *
 * Replaces EXEC_NPU_CMD(aclnnCentSelect, query, l1_cent, block_ids, block_table, seq_len, importance,
 * page_position, page_position_length, max_page_position_length)
 * (extension/csrc/add_custom.cpp:998-999; kernel argument order src/transformer/cent_select/cent_select.cpp:538).
 * `importance` is allocated uninitialised by the caller and never read by the reference kernel
 * (add_custom.cpp:995; cent_select.cpp:55,216): it may be NULL and is ignored.
 * The batch size is seqLen.shape[0], not query.shape[0] (cent_select_tiling.cpp:32,53).
 * Top-k size, sink/recent pages and the pageLen/8 cap: see spfaSetTopK / spfaSetSinkRecent / spfaSetPageCap.
 *
 * maxPagePositionLengthOut [B,8] int64.  Other shapes as in aclnn_compute_cent.h / aclnn_select_position.h.
 */
#ifndef SPFA_ACLNN_CENT_SELECT_H_
#define SPFA_ACLNN_CENT_SELECT_H_
#include "acl_shim.h"
#ifdef __cplusplus
extern "C" {
#endif
SPFA_API aclnnStatus aclnnCentSelectGetWorkspaceSize(const aclTensor* query, const aclTensor* l1Cent,
                                                     const aclTensor* blockIds, const aclTensor* blockTable,
                                                     const aclTensor* seqLen, const aclTensor* importance,
                                                     const aclTensor* pagePositionOut,
                                                     const aclTensor* pagePositionLengthOut,
                                                     const aclTensor* maxPagePositionLengthOut,
                                                     uint64_t* workspaceSize, aclOpExecutor** executor);
SPFA_API aclnnStatus aclnnCentSelect(void* workspace, uint64_t workspaceSize, aclOpExecutor* executor,
                                     aclrtStream stream);
#ifdef __cplusplus
}
#endif
#endif
