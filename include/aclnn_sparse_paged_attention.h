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
 * (per channel) or [2] (per tensor) in the query dtype, index 0 = key, 1 = value.
 * blocktable [>=B,maxBlocks] i32.
 * blockposition [B,Hq|Hkv,P] i32, a row ends at its first 0x7fffffff.  actualSeqLengths: host int64,
 * 1 or B values (NULL: the whole list).  attentionOut: query's shape; dtype of query (or float32), or int8
 * together with quantScale2.
 *
 * Optional operands of the attention core (NULL or an empty tensor = absent):
 *   pseShift   [B|1, Hq, 1, S] in the query dtype, S >= maxBlocks*blockSize: added to the scaled scores
 *              (sparse_paged_fusion_attention_tiling_check.cc:737-803; kernel …_split_Bbn2s2_Us2.h:3366-3380);
 *   attenMask  [B, ..., S] bool | int8 | uint8, one row of S >= maxBlocks*blockSize flags per batch: a non-zero
 *              flag replaces the score by -1e12 (…_tiling_check.cc:805-844; kernel :3382-3404, :860).
 *              Both index the COLUMNS of the gathered sequence: list slot * blockSize + token (:2192-2194,
 *              2239-2244).  With either one present every q head is streamed as a pass of its own;
 *   quantScale2 / quantOffset2  float32 (or bfloat16 with a bfloat16 query), 1 value or Hq*D values
 *              ([1,Hq,1,D] and the other spellings of …_tiling_check.cc:412-464): attentionOut is int8,
 *              round(out * scale2 + offset2) through float16, ties away from zero, saturating (kernel :4016-4111).
 * dequantScale1, quantScale1, dequantScale2 (int8 queries) must be NULL: not on this path.  kvPaddingSize is accepted and
 * ignored: left padding only exists for contiguous KV, with a block table the reference drops it as well
 * (sparse_paged_fusion_attention_tiling_check.cc:590-594).
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
