/*
 * SelectPosition: for every (b, q head) the ascending logical page indices i < ceil(seq_len[b]/128)
 * whose cluster id block_ids[h/G][block_table[b][i]] occurs in indices[b,h,:].
 *
 * Replaces EXEC_NPU_CMD(aclnnSelectPosition, block_ids, block_table, seq_len, indices, page_position,
 * page_position_length, block_table_gather) (extension/csrc/add_custom.cpp:968; kernel
 * src/transformer/select_position/select_position.cpp:122-193).
 *
 * blockIds [Hkv,kvPageLen] i32, blockTable [maxBatch,maxPage] i32, seqLen [B] i32, indices [B,Hq,k] i32,
 * pagePositionOut [B,Hq,P] i32 (padded with 0x7fffffff), pagePositionLengthOut [B,Hq,8] i32 (value
 * replicated 8x), blockTableGatherOut [B,Hq,maxPage] i32.  The page size is 128 (select_position.cpp:112).
 */
#ifndef SPFA_ACLNN_SELECT_POSITION_H_
#define SPFA_ACLNN_SELECT_POSITION_H_
#include "acl_shim.h"
#ifdef __cplusplus
extern "C" {
#endif
SPFA_API aclnnStatus aclnnSelectPositionGetWorkspaceSize(const aclTensor* blockIds, const aclTensor* blockTable,
                                                         const aclTensor* seqLen, const aclTensor* indices,
                                                         const aclTensor* pagePositionOut,
                                                         const aclTensor* pagePositionLengthOut,
                                                         const aclTensor* blockTableGatherOut,
                                                         uint64_t* workspaceSize, aclOpExecutor** executor);
SPFA_API aclnnStatus aclnnSelectPosition(void* workspace, uint64_t workspaceSize, aclOpExecutor* executor,
                                         aclrtStream stream);
#ifdef __cplusplus
}
#endif
#endif
