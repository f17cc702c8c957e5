/*
 * Descriptor shim for the aclnn-style two-phase C-ABI on MI355X (HIP).
 *
 * The reference's bindings describe every operand with opaque CANN descriptors before calling
 * aclnn<Op>GetWorkspaceSize / aclnn<Op> (extension/csrc/add_custom.cpp:30-118,836-895;
 * dlsym'd prototypes in extension/csrc/pytorch_npu_helper.hpp:47-72).  This header declares the
 * same entry points with the same argument order, enum values and ownership rules so those call
 * sites bind unchanged; device pointers are HIP device pointers and aclrtStream is a hipStream_t.
 *
 * Ownership (add_custom.cpp:906-929): the caller owns tensor memory and descriptors and may destroy
 * descriptors right after the exec call returns; GetWorkspaceSize therefore copies shapes, strides,
 * pointers and aclIntArray contents into the executor.  The executor is consumed by the exec call
 * unless aclSetAclOpExecutorRepeatable was called on it, in which case the caller destroys it with
 * aclDestroyAclOpExecutor (add_custom.cpp:847,929).
 */
#ifndef SPFA_ACL_SHIM_H_
#define SPFA_ACL_SHIM_H_

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPFA_API __attribute__((visibility("default")))

typedef int aclnnStatus;   /* 0 = success */
typedef void* aclrtStream; /* hipStream_t */

/* status codes (subset of CANN's aclnn error space) */
#define ACLNN_SUCCESS 0
#define ACLNN_ERR_PARAM_NULLPTR 161001
#define ACLNN_ERR_PARAM_INVALID 161002
#define ACLNN_ERR_RUNTIME_ERROR 361001
#define ACLNN_ERR_INNER 561000

/* values of CANN's aclDataType / aclFormat that the reference's binding uses
 * (add_custom.cpp:41-70, :74 ACL_FORMAT_ND) */
typedef enum {
    ACL_DT_UNDEFINED = -1,
    ACL_FLOAT = 0,
    ACL_FLOAT16 = 1,
    ACL_INT8 = 2,
    ACL_INT32 = 3,
    ACL_UINT8 = 4,
    ACL_INT16 = 6,
    ACL_UINT16 = 7,
    ACL_UINT32 = 8,
    ACL_INT64 = 9,
    ACL_UINT64 = 10,
    ACL_DOUBLE = 11,
    ACL_BOOL = 12,
    ACL_BF16 = 27
} aclDataType;

typedef enum { ACL_FORMAT_UNDEFINED = -1, ACL_FORMAT_NCHW = 0, ACL_FORMAT_NHWC = 1, ACL_FORMAT_ND = 2 } aclFormat;

typedef struct aclTensor aclTensor;
typedef struct aclTensorList aclTensorList;
typedef struct aclIntArray aclIntArray;
typedef struct aclOpExecutor aclOpExecutor;

/* pytorch_npu_helper.hpp:55-58 */
SPFA_API aclTensor* aclCreateTensor(const int64_t* viewDims, uint64_t viewDimsNum, aclDataType dataType,
                                    const int64_t* stride, int64_t offset, aclFormat format,
                                    const int64_t* storageDims, uint64_t storageDimsNum, void* tensorData);
SPFA_API aclnnStatus aclDestroyTensor(const aclTensor* tensor);
/* takes ownership of the listed tensors, like CANN (add_custom.cpp:101,907-908) */
SPFA_API aclTensorList* aclCreateTensorList(const aclTensor* const* value, uint64_t size);
SPFA_API aclnnStatus aclDestroyTensorList(const aclTensorList* list);
/* copies the host values (add_custom.cpp:113-117) */
SPFA_API aclIntArray* aclCreateIntArray(const int64_t* value, uint64_t size);
SPFA_API aclnnStatus aclDestroyIntArray(const aclIntArray* array);

SPFA_API aclnnStatus aclSetAclOpExecutorRepeatable(aclOpExecutor* executor);
SPFA_API aclnnStatus aclDestroyAclOpExecutor(aclOpExecutor* executor);
/* thread-local text of the most recent failure (add_custom.cpp:860) */
SPFA_API const char* aclGetRecentErrMsg(void);

/* Looked up but null-tolerated by the reference's dispatch macro
 * (pytorch_npu_helper.hpp:727-748,785-798); provided as no-ops. */
SPFA_API int InitHugeMemThreadLocal(void* arg, bool sync);
SPFA_API void UnInitHugeMemThreadLocal(void* arg, bool sync);
SPFA_API void ReleaseHugeMem(void* arg, bool sync);

/*
 * Extensions (no counterpart in the reference): per-thread knobs read by the next
 * aclnn<Op>GetWorkspaceSize on the calling thread.  The reference hard-codes these
 * (k: add_custom.cpp:940, cent_select_tiling.cpp:58, ..._tiling.cc:1603; sink/recent pages and the
 * pageLen/8 cap: cent_select.cpp:218,350-362,410-414).  Defaults: k = 64, sink_recent = 0,
 * cap_pages = 0 (the reference's Python/CPU path).
 */
/* The four selection knobs below are ONE-SHOT: the next aclnn<Op>GetWorkspaceSize of an op that reads them
 * (CentSelect, SelectPosition: page size only, SparsePagedFusionAttention) takes them and restores the defaults,
 * whether that call succeeds or fails.  A C caller that wants the device kernel's behaviour of the reference
 * (cent_select.cpp: k = 64, sink/recent pages, pageLen/8 cap) calls spfaSetSinkRecent(1); spfaSetPageCap(1); before
 * every phase 1. */
SPFA_API aclnnStatus spfaSetTopK(int64_t k);
SPFA_API aclnnStatus spfaSetSinkRecent(int enable);
SPFA_API aclnnStatus spfaSetPageCap(int enable);
/* Small batches (fewer items than half the workgroups the chip holds at once): the two attention ops deal each item's page list to up
 * to maxSplit workgroups and merge their states in the same launch; the reported workspace size grows by the
 * states.  1 turns the split off.  Default 16, range [1, 16]. */
SPFA_API aclnnStatus spfaSetMaxSplit(int64_t maxSplit);
/* Launches with several items per workgroup: the last item of every workgroup is streamed in chunks that
 * workgroups which ran out of work may take over (states merged in the same launch; workspace grows by one data
 * area per resident workgroup).  0: off (bit-reproducible attentionOut); 1 (default): on where the planner expects
 * it to pay (several items per workgroup, long lists, group passes); 2: on for every launch without a page-list
 * split. */
SPFA_API aclnnStatus spfaSetTailSharing(int enable);
/* Fused op, shared passes over the centroids.  With a page-list split the workgroups of one item score its rows in
 * one shared pass and exchange the score keys through the workspace (a bounded wait; a workgroup that gives up
 * scores alone).  In a launch of exactly J items per resident workgroup (one q head per pass, workgroup count a
 * multiple of the kv heads) every workgroup owns J sequences of one kv head and scores their J rows in one pass
 * before it streams the first.  Outputs do not depend on the setting.  0 turns both off.  Default on. */
SPFA_API aclnnStatus spfaSetSharedScoring(int enable);
/* Workgroup geometry of the one-q-head attention kernels: 1 = one workgroup of eight streamer waves per CU, 2 = two
 * workgroups of four, 0 (default) = the planner's choice (eight, unless a fused launch hands a workgroup more items
 * than can be scored ahead).  Outputs do not depend on it beyond fp32 rounding.  Stays until changed. */
SPFA_API aclnnStatus spfaSetWorkgroupGeometry(int geometry);
/* Fused op, one q head per pass, more items than resident workgroups: the work queue hands out groups of up to 8
 * consecutive sequences of one kv head and their rows are scored in one pass over the head's centroids, beside the page
 * stream of the group before (one pass per item otherwise).  Outputs do not depend on the setting.  0 turns it off.
 * Default on.  Stays until changed. */
SPFA_API aclnnStatus spfaSetBatchedSelection(int enable);
/* Group passes (several q heads share the K / V rows, matrix-core streamer), head dims up to 128: K rows are fetched
 * whole — every request a full cache line — and turned into matrix operand order through a wave-private LDS image, as V
 * rows are (the image lies in the bytes of the tile's V image: no LDS of its own).  0: strided K loads straight into
 * operand order, as at head dim 256.  Outputs do not depend on the setting (the same products in the same order).
 * Default on.  Stays until changed. */
SPFA_API aclnnStatus spfaSetKThroughLds(int enable);
/* page size of the two ops without a blockSize attribute (CentSelect, SelectPosition); the reference
 * compiles in 128 (cent_select.cpp:17, select_position.cpp:112).  Default 128. */
SPFA_API aclnnStatus spfaSetPageSize(int64_t pageSize);
/* One-shot: the next aclnnSparsePagedAttentionGetWorkspaceSize on this thread reads actual_seq_lengths
 * from this DEVICE array of n int64 instead of the host aclIntArray (the reference's binding copies a
 * device tensor to the host first, add_custom.cpp:105-118). */
SPFA_API aclnnStatus spfaSetActualSeqLengthsDevice(const int64_t* deviceLengths, int64_t n);
/* One-shot: the next aclnnSparsePagedAttentionGetWorkspaceSize on this thread reads `blockposition` [B, numHeads |
 * numKeyValueHeads, maxTokens] as LOGICAL TOKEN indices instead of page indices (token t = row t % blockSize of page
 * blocktable[b][t / blockSize]; a row ends at its first 0x7fffffff) and actualSeqLengths as token counts — the
 * token-level variant the reference sketches in tests/ifa_v5_case_tkp.py:53-54.  Head dims 64 and 128; pseShift /
 * attenMask are refused; any maxTokens (lists above 4096 entries are streamed in windows).  With numHeads rows every q
 * head is a pass of its own; with numKeyValueHeads rows a list is streamed once for the q heads of its group. */
SPFA_API aclnnStatus spfaSetTokenPositions(int enable);
/* One-shot: antiquant mode of the next phase 1 of the two sparse attention ops on this thread — the antiquantMode
 * attribute of the kernel the sparse ops were derived from (sparse_paged_fusion_attention_tiling.cc:719-765,982-984),
 * which their aclnn signatures do not carry.  0 (default): antiquantScale / antiquantOffset are [2, Hkv, D] (per channel)
 * or [2] (per tensor) in the query dtype.  1: per token — float32 [2, B, S] with S >= blocktable columns * blockSize
 * (+ the shared prefix length), key parameters first: K_t = (k_t + offset[0, b, c]) * scale[0, b, c] and likewise V with
 * index 1, c being the token's column in the gathered sequence (prefix tokens, then list slot * blockSize + token), as
 * for pseShift / attenMask (sparse_paged_fusion_attention_split_Bbn2s2_Us2.h:2039-2043, 2262-2266, 3711-3822).
 * 2: per tensor and kv head — [2, Hkv] in the query dtype, one key and one value parameter per kv head
 * (sparse_paged_fusion_attention_tiling_check.cc:488-497; the reference takes key and value parameters as two [Hkv]
 * tensors, here they are the two rows of one).  3: per token and kv head — float32 [2, B, Hkv, S] (:466-487, kernel
 * offsets …Us2.h:2203-2206).  Modes 4 and 5 of the reference (parameters in page-attention layout) are refused. */
SPFA_API aclnnStatus spfaSetAntiquantMode(int64_t mode);
/* One-shot: the next aclnnSparsePagedAttentionGetWorkspaceSize / aclnnSparsePagedFusionAttentionGetWorkspaceSize on
 * this thread also writes the softmax log-sum-exp of every output row, ln(sum_t exp(score_t)) as float32, to this
 * DEVICE array of n >= B * numHeads values laid out [B, numHeads] — the softmaxLse output of the reference kernel
 * (sparse_paged_fusion_attention_split_Bbn2s2_Us2.h:3130-3159; a row that attends nothing gets -3.4e38, :1289-1300),
 * for which the aclnn signatures of the two sparse ops have no operand.  NULL disarms. */
SPFA_API aclnnStatus spfaSetSoftmaxLseOut(float* deviceLse, int64_t n);
/* One-shot: the next phase 1 of the two sparse attention ops on this thread lets every output row attend, in front of
 * its gathered pages, the first actualPrefixLen tokens of a prefix shared by the whole batch — the keySharedPrefix /
 * valueSharedPrefix / actualSharedPrefixLen operands of the kernel the sparse ops were derived from
 * (sparse_paged_fusion_attention_split_Bbn2s2_Us2.h:1418-1439; their aclnn signatures carry no such operands).
 * The two DEVICE tensors have the KV cache's dtype and layout with batch 1: [1, prefixTokens, Hkv * D] for a
 * [blocks, blockSize, Hkv * D] cache, [1, Hkv, prefixTokens, D] for a [blocks, Hkv, blockSize, D] cache; int8
 * prefixes use the op's antiquant parameters.  pseShift / attenMask columns then start with the prefix tokens
 * (:2263-2266).  Not available with token-level positions.  NULL pointers disarm. */
SPFA_API aclnnStatus spfaSetSharedPrefix(const void* keyPrefix, const void* valuePrefix, int64_t prefixTokens,
                                         int64_t actualPrefixLen);
/* A launch of the two attention ops starts with a small kernel that zeroes the counters at the head of the workspace;
 * the op's kernel leaves them zeroed again when it ends.  enable = 1 on a repeatable executor of
 * SparsePagedFusionAttention promises that nothing but this executor's launches writes the workspace between
 * them: the zeroing kernel then runs only the first time a workspace pointer is seen, and every later call is exactly
 * one kernel launch.  Default 0 (the workspace is scratch that the caller may reuse between calls). */
SPFA_API aclnnStatus spfaSetWorkspacePersistent(aclOpExecutor* executor, int enable);
/* Where the counters of a launch of the two attention ops live (per thread, stays until changed; default 1).
 * 1: when they fit 64 KiB (batches up to about 4,000 sequences) they are kept in a slot of device memory that the
 * library owns — one slot per (device, stream), sixteen per device, zeroed once and left zeroed by every launch — so
 * a call with any caller workspace enqueues exactly one kernel and only the data areas of the workspace are used.
 * Launches on one stream are ordered, which is what makes sharing a slot safe.  Launches that are being captured into
 * a graph (a replay may run on any stream), launches on the hipStreamPerThread handle and launches that find all sixteen
 * slots owned by streams with work in flight keep their counters in the caller's workspace behind the zeroing kernel, as
 * with 0.  CentSelect keeps its per-sequence maxima the same way (one kernel instead of two). */
SPFA_API aclnnStatus spfaSetLibraryCounters(int enable);
/* The launch plan of an executor of the two attention ops as one line of text (kernel form, workgroup geometry, rows per
 * pass, items scored ahead, super-items, page-list split, sharing of last items, resident workgroups, LDS and workspace
 * bytes, entries of a token-level list held in LDS at a time): for logs and tests.  Returns the number of characters written, -1 on a bad argument. */
SPFA_API int spfaDescribeExecutor(const aclOpExecutor* executor, char* buf, uint64_t n);
/* Kernels the library has enqueued (or recorded into a graph) since it was loaded, over all threads and devices: the
 * difference around a call tells how many kernels that call was. */
SPFA_API uint64_t spfaKernelLaunches(void);
SPFA_API const char* spfaVersion(void);

#ifdef __cplusplus
}
#endif
#endif /* SPFA_ACL_SHIM_H_ */
