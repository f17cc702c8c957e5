/*
 * AssignBlockIds: cluster ids for newly filled KV pages — the producer of block_ids.
 *
 * No counterpart in the reference: its ops only consume block_ids [Hkv, kvPageLen] ("cluster id of every physical
 * page", README.md:33) and l1_cent; nothing in the repository computes them.  A decode loop needs the step, so it is
 * provided behind the same two-phase ABI:
 *
 * for every sequence b < seqLen.shape[0], every logical page i with
 *     firstNewPage[b] <= i < ceil(seqLen[b] / blockSize)          (at most maxNewPages of them per call)
 * and every kv head n:
 *     m  = mean over the page's live tokens of K[blockTable[b][i], token, n, :]   (fp32; int8 K is dequantised with the
 *          key parameters of antiquantScale / antiquantOffset, the layout of the attention ops)
 *     m' = m rounded to l1Cent's dtype
 *     blockIdsInout[n, blockTable[b][i]] = argmax_c  m' . l1Cent[n, c, :]         (fp32 accumulate; ties: smallest c)
 * Pages outside that range keep their ids.  The caller advances firstNewPage itself.
 *
 * key: one-tensor list holding the paged K cache, [blocks, blockSize, Hkv*D], [blocks, blockSize, Hkv, D] or
 * [blocks, Hkv, blockSize, D]; f16 / bf16 (same dtype as l1Cent) or int8.  blockTable [>=B, maxPage] i32, seqLen [B] i32,
 * firstNewPage [B] i32, l1Cent [Hkv, C, D] f16 | bf16, blockIdsInout [Hkv, kvPageLen] i32.  maxNewPages <= 0 means
 * "all pages of the block table".  Workspace size is 0.
 */
#ifndef SPFA_ACLNN_ASSIGN_BLOCK_IDS_H_
#define SPFA_ACLNN_ASSIGN_BLOCK_IDS_H_
#include "acl_shim.h"
#ifdef __cplusplus
extern "C" {
#endif
SPFA_API aclnnStatus aclnnAssignBlockIdsGetWorkspaceSize(const aclTensorList* key, const aclTensor* blockTable,
                                                         const aclTensor* seqLen, const aclTensor* firstNewPage,
                                                         const aclTensor* l1Cent, const aclTensor* antiquantScale,
                                                         const aclTensor* antiquantOffset, int64_t blockSize,
                                                         int64_t maxNewPages, const aclTensor* blockIdsInout,
                                                         uint64_t* workspaceSize, aclOpExecutor** executor);
SPFA_API aclnnStatus aclnnAssignBlockIds(void* workspace, uint64_t workspaceSize, aclOpExecutor* executor,
                                         aclrtStream stream);
#ifdef __cplusplus
}
#endif
#endif
