/*
 * SparsePagedFusionAttention: CentSelect then SparsePagedAttention in ONE kernel launch; the page
 * lists never return to the host.  Each q head attends over its own page list; block_position,
 * page_position_length and max_page_position_length are also written out.
 *
 * Same prototype as
 * src/transformer/sparse_paged_fusion_attention/ophost/aclnn_sparse_paged_fusion_attention.h:60-81
 * (attrs BEFORE the three aux outputs; call site extension/csrc/add_custom.cpp:836-844,895).
 *
 * l1_cent [Hkv,C,D] query dtype, block_ids [Hkv,kvPageLen] i32, total_seq_len [B] i32,
 * block_position [B,Hq,P] i32 out, page_position_length [B,Hq,8] i32 out,
 * max_page_position_length [B,8] i64 out (any of the three may be NULL).  actualSeqLengths is a host-side
 * tiling hint in the reference (the device recomputes the lengths, sparse_paged_fusion_attention.cpp:68)
 * and is ignored here.  Other operands as in aclnn_sparse_paged_attention.h.
 */
#ifndef SPFA_ACLNN_SPARSE_PAGED_FUSION_ATTENTION_H_
#define SPFA_ACLNN_SPARSE_PAGED_FUSION_ATTENTION_H_
#include "acl_shim.h"
#ifdef __cplusplus
extern "C" {
#endif
SPFA_API aclnnStatus aclnnSparsePagedFusionAttentionGetWorkspaceSize(
    const aclTensor* query, const aclTensorList* key, const aclTensorList* value, const aclTensor* pseShift,
    const aclTensor* attenMask, const aclIntArray* actualSeqLengths, const aclTensor* dequantScale1,
    const aclTensor* quantScale1, const aclTensor* dequantScale2, const aclTensor* quantScale2,
    const aclTensor* quantOffset2, const aclTensor* antiquantScale, const aclTensor* antiquantOffset,
    const aclTensor* blocktable, const aclTensor* kvPaddingSize, const aclTensor* l1_cent,
    const aclTensor* block_ids, const aclTensor* total_seq_len, int64_t numHeads, double scaleValue,
    char* inputLayout, int64_t numKeyValueHeads, int64_t blockSize, int64_t innerPrecise,
    const aclTensor* block_position, const aclTensor* page_position_length,
    const aclTensor* max_page_position_length, const aclTensor* attentionOut, uint64_t* workspaceSize,
    aclOpExecutor** executor);
SPFA_API aclnnStatus aclnnSparsePagedFusionAttention(void* workspace, uint64_t workspaceSize,
                                                     aclOpExecutor* executor, const aclrtStream stream);
/*
 * No counterpart in the reference.  For callers that cannot allocate device memory: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  Phase 2 of the fused op without a workspace argument: the
 * counters, boards and partial states that aclnnSparsePagedFusionAttention keeps in the caller's workspace live in
 * a fixed 32 MiB array inside the library (one copy per GPU, held until the process ends).  The executor comes from
 * aclnnSparsePagedFusionAttentionGetWorkspaceSize as usual (inputs and outputs are bound there); a plan whose
 * reported workspace exceeds 32 MiB is rejected with ACLNN_ERR_PARAM_INVALID and a text that says so.  The call
 * allocates and frees nothing and enqueues on `stream` only.  All calls on one GPU share the array: calls that
 * overlap on one GPU (two streams, two threads) race; calls ordered on one stream are safe.
 */
SPFA_API aclnnStatus aclnnSparsePagedFusionAttentionNoAlloc(aclOpExecutor* executor, const aclrtStream stream);
#ifdef __cplusplus
}
#endif
#endif
