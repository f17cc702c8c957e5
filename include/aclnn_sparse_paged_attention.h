/*
 * SparsePagedAttention: decode attention over a virtual sequence made of selected pages.
 * Virtual token t of batch b lives in virtual page j = t / blockSize, logical page
 * p = blockposition[b, r, j], physical block blocktable[b, p].
 *
 * Same prototype as src/transformer/sparse_paged_attention/ophost/aclnn_sparse_paged_attention.h:50-69
 * (call site extension/csrc/add_custom.cpp sparse_paged_attention_impl_npu).
 *
 * query [B,Hq,D] | [B,Hq,1,D] | [B,1,Hq*D], f16|bf16.  key/value: one-tensor lists holding the paged
 * cache, [blocks,bs,Hkv*D] / [blocks,bs,Hkv,D] (BSH) or [blocks,Hkv,bs,D] (BNSD); dtype of query, or
 * int8 together with antiquantScale (+ optional antiquantOffset), each [2,Hkv,1,D] | [2,Hkv,D] | [2,Hkv*D]
 * in the query dtype, index 0 = key, 1 = value.  blocktable [>=B,maxBlocks] i32.
 * blockposition [B,Hq|Hkv,P] i32, a row ends at its first 0x7fffffff.  actualSeqLengths: host int64,
 * 1 or B values (NULL: the whole list).  attentionOut: query's shape; dtype of query (or float32).
 * pseShift, attenMask, dequant/quantScale*, quantOffset2, kvPaddingSize must be NULL (not on this path).
 */
#ifndef SPFA_ACLNN_SPARSE_PAGED_ATTENTION_H_
#define SPFA_ACLNN_SPARSE_PAGED_ATTENTION_H_
#include "acl_shim.h"
#ifdef __cplusplus
extern "C" {
#endif
SPFA_API aclnnStatus aclnnSparsePagedAttentionGetWorkspaceSize(
    const aclTensor* query, const aclTensorList* key, const aclTensorList* value, const aclTensor* pseShift,
    const aclTensor* attenMask, const aclIntArray* actualSeqLengths, const aclTensor* dequantScale1,
    const aclTensor* quantScale1, const aclTensor* dequantScale2, const aclTensor* quantScale2,
    const aclTensor* quantOffset2, const aclTensor* antiquantScale, const aclTensor* antiquantOffset,
    const aclTensor* blocktable, const aclTensor* kvPaddingSize, const aclTensor* blockposition, int64_t numHeads,
    double scaleValue, char* inputLayout, int64_t numKeyValueHeads, int64_t blockSize, int64_t innerPrecise,
    const aclTensor* attentionOut, uint64_t* workspaceSize, aclOpExecutor** executor);
SPFA_API aclnnStatus aclnnSparsePagedAttention(void* workspace, uint64_t workspaceSize, aclOpExecutor* executor,
                                               const aclrtStream stream);
#ifdef __cplusplus
}
#endif
#endif
