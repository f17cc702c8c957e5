/*
 * ComputeCent: indices[b,h,:] = ids of the k largest q[b,h,:] . l1_cent[h/G,c,:], descending,
 * equal scores ordered by ascending id.  k = indicesOut.shape[2] (compute_cent_tiling.cpp:45).
 *
 * Replaces the live call site EXEC_NPU_CMD(aclnnComputeCent, query, l1_cent, result)
 * (extension/csrc/add_custom.cpp:946; README.md:129-131).  The 8- and 10-tensor prototypes in
 * src/transformer/compute_cent/ophost/aclnn_compute_cent.h:36-39 and misc/aclnn_compute_cent.h:31-43
 * are stale generations of the same op and are not provided.
 *
 * query [B,Hq,D] f16|bf16, l1Cent [Hkv,C,D] same dtype, indicesOut [B,Hq,k] int32.
 * D in {32, 64, 128, 256}, k <= min(C, 1024).
 */
#ifndef SPFA_ACLNN_COMPUTE_CENT_H_
#define SPFA_ACLNN_COMPUTE_CENT_H_
#include "acl_shim.h"
#ifdef __cplusplus
extern "C" {
#endif
SPFA_API aclnnStatus aclnnComputeCentGetWorkspaceSize(const aclTensor* query, const aclTensor* l1Cent,
                                                      const aclTensor* indicesOut, uint64_t* workspaceSize,
                                                      aclOpExecutor** executor);
SPFA_API aclnnStatus aclnnComputeCent(void* workspace, uint64_t workspaceSize, aclOpExecutor* executor,
                                      aclrtStream stream);
#ifdef __cplusplus
}
#endif
#endif
