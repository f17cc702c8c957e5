// A compiled caller of the aclnn-style C-ABI, written against include/*.h only (plus the HIP runtime for device memory):
// the call sequences of the reference's own C++ callers, to prove that such call sites bind and run unchanged.
//
//   * fused op: extension/csrc/add_custom.cpp:770-935 — one descriptor per operand (sizes + strides, NULL for an
//     absent operand), one-tensor lists for key / value, an aclIntArray for actual_seq_lengths,
//     aclnnSparsePagedFusionAttentionGetWorkspaceSize (:836-844), aclSetAclOpExecutorRepeatable (:847), a workspace of
//     the reported size (:866-873), the launch repeated on one executor (:886-895; the reference repeats it 101 times,
//     twice is enough here), every descriptor destroyed (:906-927), aclDestroyAclOpExecutor (:929);
//   * compute_cent: the create -> GetWorkspaceSize -> exec -> destroy flow of
//     src/transformer/compute_cent/test/test_compute_cent.cpp:150-260 with the live three-tensor signature
//     (extension/csrc/add_custom.cpp:946).
//
// usage: ref_caller <dir>          reads <dir>/meta.txt and the *.bin inputs, writes the *.bin outputs
//        ref_caller --abi-check    no GPU: builds descriptors and runs a phase 1 on fake addresses
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "aclnn_cent_select.h"
#include "aclnn_compute_cent.h"
#include "aclnn_sparse_paged_fusion_attention.h"

#define HIP_OK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            fprintf(stderr, "%s failed: %s\n", #expr, hipGetErrorString(e_));                \
            return 2;                                                                        \
        }                                                                                    \
    } while (0)
#define ACL_OK(expr)                                                                         \
    do {                                                                                     \
        int r_ = (expr);                                                                     \
        if (r_ != 0) {                                                                       \
            fprintf(stderr, "%s failed: %d: %s\n", #expr, r_, aclGetRecentErrMsg());         \
            return 3;                                                                        \
        }                                                                                    \
    } while (0)

namespace {

struct DeviceTensor {            // what at::Tensor is to the reference's binding: shape, dtype, device memory
    std::vector<int64_t> sizes, strides;
    aclDataType dtype;
    void* data = nullptr;
    size_t bytes = 0;
};

size_t elem_bytes(aclDataType t) {
    switch (t) {
        case ACL_FLOAT16: case ACL_BF16: return 2;
        case ACL_INT8: return 1;
        case ACL_INT64: return 8;
        default: return 4;
    }
}

DeviceTensor make(std::vector<int64_t> sizes, aclDataType dtype) {
    DeviceTensor t;
    t.sizes = sizes;
    t.dtype = dtype;
    t.strides.assign(sizes.size(), 1);
    for (int i = (int)sizes.size() - 2; i >= 0; --i) t.strides[i] = t.strides[i + 1] * sizes[i + 1];
    size_t n = 1;
    for (int64_t s : sizes) n *= (size_t)s;
    t.bytes = n * elem_bytes(dtype);
    return t;
}

// add_custom.cpp:29-76
aclTensor* ConvertTensorToAcl(const DeviceTensor& t) {
    if (t.bytes == 0) return nullptr;
    return aclCreateTensor(t.sizes.data(), t.sizes.size(), t.dtype, t.strides.data(), 0, ACL_FORMAT_ND, t.sizes.data(),
                           t.sizes.size(), t.data);
}
// add_custom.cpp:79-102
aclTensorList* ConvertTensorListToAcl(const std::vector<const DeviceTensor*>& list) {
    std::vector<aclTensor*> acl;
    for (const DeviceTensor* t : list)
        if (aclTensor* a = ConvertTensorToAcl(*t)) acl.push_back(a);
    if (acl.empty()) return nullptr;
    return aclCreateTensorList(acl.data(), acl.size());
}

bool read_file(const std::string& path, std::vector<char>& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return true;
}

int upload(const std::string& dir, const char* name, DeviceTensor& t) {
    std::vector<char> host;
    if (!read_file(dir + "/" + name, host) || host.size() != t.bytes) {
        fprintf(stderr, "%s: expected %zu bytes, found %zu\n", name, t.bytes, host.size());
        return 4;
    }
    HIP_OK(hipMalloc(&t.data, t.bytes));
    HIP_OK(hipMemcpy(t.data, host.data(), t.bytes, hipMemcpyHostToDevice));
    return 0;
}

int download(const std::string& dir, const char* name, const DeviceTensor& t) {
    std::vector<char> host(t.bytes);
    HIP_OK(hipMemcpy(host.data(), t.data, t.bytes, hipMemcpyDeviceToHost));
    std::ofstream f(dir + "/" + name, std::ios::binary);
    f.write(host.data(), (std::streamsize)host.size());
    return f ? 0 : 4;
}

int abi_check() {
    // phase 1 is host code: it can be exercised without a GPU on addresses that are never dereferenced
    DeviceTensor q = make({2, 8, 64}, ACL_FLOAT16), l1 = make({2, 256, 64}, ACL_FLOAT16), idx = make({2, 8, 8}, ACL_INT32);
    q.data = l1.data = idx.data = (void*)0x10000;
    aclTensor *qt = ConvertTensorToAcl(q), *lt = ConvertTensorToAcl(l1), *it = ConvertTensorToAcl(idx);
    uint64_t ws = 1;
    aclOpExecutor* ex = nullptr;
    ACL_OK(aclnnComputeCentGetWorkspaceSize(qt, lt, it, &ws, &ex));
    if (ws != 0 || ex == nullptr) return 5;
    aclDestroyAclOpExecutor(ex);
    int r = aclnnComputeCentGetWorkspaceSize(qt, nullptr, it, &ws, &ex);
    if (r != ACLNN_ERR_PARAM_NULLPTR || strstr(aclGetRecentErrMsg(), "l1_cent") == nullptr) return 6;
    aclDestroyTensor(qt);
    aclDestroyTensor(lt);
    aclDestroyTensor(it);
    printf("abi ok: %s\n", spfaVersion());
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc == 2 && strcmp(argv[1], "--abi-check") == 0) return abi_check();
    if (argc != 2) {
        fprintf(stderr, "usage: %s <dir> | --abi-check\n", argv[0]);
        return 1;
    }
    const std::string dir = argv[1];
    std::map<std::string, long long> m;
    {
        std::ifstream f(dir + "/meta.txt");
        std::string key;
        long long val;
        while (f >> key >> val) m[key] = val;
    }
    const int64_t B = m["B"], Hq = m["Hq"], Hkv = m["Hkv"], D = m["D"], C = m["C"], k = m["k"], bs = m["bs"],
                  max_page = m["max_page"], n_blocks = m["n_blocks"], kv_page_len = m["kv_page_len"], P = m["P"];
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || D <= 0) {
        fprintf(stderr, "meta.txt is incomplete\n");
        return 1;
    }

    DeviceTensor query = make({B, Hq, D}, ACL_FLOAT16), l1_cent = make({Hkv, C, D}, ACL_FLOAT16);
    DeviceTensor key = make({n_blocks, bs, Hkv, D}, ACL_FLOAT16), value = make({n_blocks, bs, Hkv, D}, ACL_FLOAT16);
    DeviceTensor blocktable = make({B, max_page}, ACL_INT32), block_ids = make({Hkv, kv_page_len}, ACL_INT32);
    DeviceTensor total_seq_len = make({B}, ACL_INT32);
    int rc;
    if ((rc = upload(dir, "q.bin", query)) || (rc = upload(dir, "l1.bin", l1_cent)) || (rc = upload(dir, "key.bin", key)) ||
        (rc = upload(dir, "val.bin", value)) || (rc = upload(dir, "table.bin", blocktable)) ||
        (rc = upload(dir, "ids.bin", block_ids)) || (rc = upload(dir, "seq.bin", total_seq_len)))
        return rc;
    hipStream_t stream = nullptr;
    HIP_OK(hipStreamCreate(&stream));

    // ---- compute_cent: create -> GetWorkspaceSize -> exec -> destroy (test_compute_cent.cpp:150-260) ----
    DeviceTensor indices = make({B, Hq, k}, ACL_INT32);
    HIP_OK(hipMalloc(&indices.data, indices.bytes));
    {
        aclTensor *q = ConvertTensorToAcl(query), *l1 = ConvertTensorToAcl(l1_cent), *idx = ConvertTensorToAcl(indices);
        uint64_t workspaceSize = 0;
        aclOpExecutor* executor = nullptr;
        ACL_OK(aclnnComputeCentGetWorkspaceSize(q, l1, idx, &workspaceSize, &executor));
        void* workspace = nullptr;
        if (workspaceSize > 0) HIP_OK(hipMalloc(&workspace, workspaceSize));
        ACL_OK(aclnnComputeCent(workspace, workspaceSize, executor, stream));   // consumes the one-shot executor
        aclDestroyTensor(q);
        aclDestroyTensor(l1);
        aclDestroyTensor(idx);
        HIP_OK(hipStreamSynchronize(stream));
        if (workspace) HIP_OK(hipFree(workspace));
    }

    // ---- fused op: add_custom.cpp:770-935 ----
    DeviceTensor attention_out = make({B, Hq, D}, ACL_FLOAT16);
    DeviceTensor block_position = make({B, Hq, P}, ACL_INT32), page_position_length = make({B, Hq, 8}, ACL_INT32);
    DeviceTensor max_page_position_length = make({B, 8}, ACL_INT64);
    for (DeviceTensor* t : {&attention_out, &block_position, &page_position_length, &max_page_position_length}) {
        HIP_OK(hipMalloc(&t->data, t->bytes));
        HIP_OK(hipMemset(t->data, 0xEE, t->bytes));
    }
    DeviceTensor empty = make({0}, ACL_FLOAT16);     // "optional" operands arrive as empty tensors (:32-35)
    aclTensor* queryTensor = ConvertTensorToAcl(query);
    aclTensorList* tensorKeyList = ConvertTensorListToAcl({&key});
    aclTensorList* tensorValueList = ConvertTensorListToAcl({&value});
    aclTensor* pseShiftTensor = ConvertTensorToAcl(empty);
    aclTensor* attenTensor = ConvertTensorToAcl(empty);
    std::vector<int64_t> hint(1, (int64_t)(max_page * bs / 8));   // the tests pass the 1/8-sparsity estimate (fused_ifa_v5_case.py:86)
    aclIntArray* actualSeqLengths = aclCreateIntArray(hint.data(), hint.size());
    aclTensor *dequantScale1Tensor = ConvertTensorToAcl(empty), *quantScale1Tensor = ConvertTensorToAcl(empty),
              *dequantScale2Tensor = ConvertTensorToAcl(empty), *quantScale2Tensor = ConvertTensorToAcl(empty),
              *quantOffset2Tensor = ConvertTensorToAcl(empty), *antiquantScaleTensor = ConvertTensorToAcl(empty),
              *antiquantOffsetTensor = ConvertTensorToAcl(empty);
    aclTensor* blocktableTensor = ConvertTensorToAcl(blocktable);
    aclTensor* kvPaddingSizeTensor = ConvertTensorToAcl(empty);
    aclTensor *l1CentTensor = ConvertTensorToAcl(l1_cent), *blockIdsTensor = ConvertTensorToAcl(block_ids),
              *totalSeqLenTensor = ConvertTensorToAcl(total_seq_len);
    aclTensor *blockPositionTensor = ConvertTensorToAcl(block_position),
              *pagePositionLengthTensor = ConvertTensorToAcl(page_position_length),
              *maxPagePositionLengthTensor = ConvertTensorToAcl(max_page_position_length),
              *attentionOutTensor = ConvertTensorToAcl(attention_out);
    char layerOut[] = "BNSD";
    const int64_t num_heads = Hq, num_key_value_heads = Hkv, block_size = bs, inner_precise = 1;
    const double scale_value = 1.0 / std::sqrt((double)D);

    ACL_OK(spfaSetTopK(k));   // the reference compiles its k in (cent_select_tiling.cpp:58); here it is a setting
    uint64_t workspaceSize = 0;
    aclOpExecutor* executor = nullptr;
    ACL_OK(aclnnSparsePagedFusionAttentionGetWorkspaceSize(
        queryTensor, tensorKeyList, tensorValueList, pseShiftTensor, attenTensor, actualSeqLengths, dequantScale1Tensor,
        quantScale1Tensor, dequantScale2Tensor, quantScale2Tensor, quantOffset2Tensor, antiquantScaleTensor,
        antiquantOffsetTensor, blocktableTensor, kvPaddingSizeTensor, l1CentTensor, blockIdsTensor, totalSeqLenTensor,
        num_heads, scale_value, layerOut, num_key_value_heads, block_size, inner_precise, blockPositionTensor,
        pagePositionLengthTensor, maxPagePositionLengthTensor, attentionOutTensor, &workspaceSize, &executor));
    ACL_OK(aclSetAclOpExecutorRepeatable(executor));
    void* workspaceAddr = nullptr;
    if (workspaceSize > 0) {
        HIP_OK(hipMalloc(&workspaceAddr, workspaceSize));
        HIP_OK(hipMemset(workspaceAddr, 0xEE, workspaceSize));   // a workspace is scratch: nothing may rely on its contents
    }
    // the launch repeated on one executor and one workspace (:886-895); the library's own count tells how many kernels
    // that was: one per launch, plus at most one that zeroes the stream's counters the first time
    const int kLaunches = 5;
    const uint64_t launched0 = spfaKernelLaunches();
    for (int i = 0; i < kLaunches; ++i) ACL_OK(aclnnSparsePagedFusionAttention(workspaceAddr, workspaceSize, executor, stream));
    const uint64_t launched = spfaKernelLaunches() - launched0;

    if (queryTensor) aclDestroyTensor(queryTensor);
    if (tensorKeyList) aclDestroyTensorList(tensorKeyList);
    if (tensorValueList) aclDestroyTensorList(tensorValueList);
    if (actualSeqLengths) aclDestroyIntArray(actualSeqLengths);
    for (aclTensor* t : {pseShiftTensor, attenTensor, dequantScale1Tensor, quantScale1Tensor, dequantScale2Tensor,
                         quantScale2Tensor, quantOffset2Tensor, antiquantScaleTensor, antiquantOffsetTensor,
                         blocktableTensor, kvPaddingSizeTensor, l1CentTensor, blockIdsTensor, totalSeqLenTensor,
                         blockPositionTensor, pagePositionLengthTensor, maxPagePositionLengthTensor, attentionOutTensor})
        if (t) aclDestroyTensor(t);
    ACL_OK(aclDestroyAclOpExecutor(executor));
    HIP_OK(hipStreamSynchronize(stream));

    if ((rc = download(dir, "indices.bin", indices)) || (rc = download(dir, "out.bin", attention_out)) ||
        (rc = download(dir, "pos.bin", block_position)) || (rc = download(dir, "plen.bin", page_position_length)) ||
        (rc = download(dir, "pmax.bin", max_page_position_length)))
        return rc;
    printf("ok: workspace %llu bytes, %llu kernels for %d launches\n", (unsigned long long)workspaceSize,
           (unsigned long long)launched, kLaunches);
    return 0;
}
