// Host side of the aclnn-style two-phase C-ABI (see include/*.h): descriptor shim, operand
// validation, launch planning (the counterpart of the reference's host tiling functions,
// cent_select_tiling.cpp:11-115 and sparse_paged_fusion_attention_tiling.cc:66-2200) and the
// stream-ordered launch.  Phase 1 (GetWorkspaceSize) is synchronous host code and copies everything
// it needs out of the descriptors; phase 2 only enqueues work on the caller's hipStream_t and never
// synchronises or allocates.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/aclnn_assign_block_ids.h"
#include "../../include/aclnn_cent_select.h"
#include "../../include/aclnn_compute_cent.h"
#include "../../include/aclnn_select_position.h"
#include "../../include/aclnn_sparse_paged_attention.h"
#include "../../include/aclnn_sparse_paged_fusion_attention.h"
#include "spfa_params.h"

struct aclTensor {
    std::vector<int64_t> dims, strides;
    int64_t offset = 0;
    aclDataType dtype = ACL_DT_UNDEFINED;
    void* data = nullptr;
};
struct aclTensorList { std::vector<aclTensor*> items; };
struct aclIntArray { std::vector<int64_t> v; };

enum OpKind { kOpComputeCent, kOpSelectPosition, kOpCentSelect, kOpAttention, kOpFusion, kOpAssign };

struct aclOpExecutor {
    OpKind op;
    bool repeatable = false;
    spfa::CentParams cent{};
    spfa::SelectParams sel{};
    spfa::AttnParams attn{};
    spfa::AssignParams assign{};
    std::vector<long long> act_seq;   // host copy of actualSeqLengths, expanded to B values
    // workspace layout of the two attention ops: [work counters | split arrival counters] (zeroed per launch),
    // split states, per-batch actualSeqLengths
    uint64_t ws_zero_bytes = 0, ws_parts_off = 0, ws_lengths_off = 0, ws_boards_off = 0, ws_tail_off = 0;
    uint64_t ws_pgdone_off = 0, ws_pgkeys_off = 0, ws_batch_off = 0;
    uint64_t ws_bytes = 0;
    int device = -1;                  // device that was current in phase 1 (-1: none could be queried)
    // A workspace that only this executor's launches write stays in the state every launch leaves it in (counters,
    // arrival words and boards zeroed by the last workgroup to exit): the zeroing kernel in front of the launch is
    // then needed only the first time a workspace is seen.  Opt-in (spfaSetWorkspacePersistent), repeatable executors.
    bool persistent_ws = false;
    void* clean_ws = nullptr;
};

namespace {

thread_local std::string g_err;
thread_local int64_t g_topk = 64;
thread_local int g_sink_recent = 0;
thread_local int g_page_cap = 0;
thread_local int g_max_split = 16;
thread_local int g_tail_sharing = 1;
thread_local int g_shared_scoring = 1;
thread_local int g_stream_slots = 1;       // 1: launches may keep their counters in library-owned per-stream slots
thread_local int g_super_items = 1;        // 1: the planner may batch the selection of queue-fed launches (super-items)
thread_local int g_k_through_lds = 1;      // 1: the matrix-core streamer takes K through LDS where the image fits
thread_local int g_geometry = 0;           // 0: planner's choice, 1: wide, 2: narrow workgroups of the one-q-head kernels
thread_local int64_t g_page_size = 128;   // PAGESIZE of cent_select.cpp:17 / select_position.cpp:112
thread_local const long long* g_act_seq_dev = nullptr;   // one-shot, consumed by the next attention plan
thread_local int64_t g_act_seq_dev_n = 0;
thread_local int g_token_positions = 0;      // one-shot: the next attention plan reads blockposition as token indices
thread_local float* g_lse_out = nullptr;     // one-shot: the next attention plan also writes the rows' log-sum-exp there
thread_local int64_t g_lse_out_n = 0;
struct SharedPrefix { const void* k; const void* v; int64_t tokens, len; };
thread_local SharedPrefix g_prefix{nullptr, nullptr, 0, 0};   // one-shot: consumed by the next attention plan
thread_local int g_antiq_mode = 0;          // one-shot: 0 per channel / per tensor, 1 per token
thread_local void* g_debug_buf = nullptr;   // diagnostic stamp buffer (see spfaSetDebugBuffer)

constexpr size_t kMaxLds = 160 * 1024;    // LDS per CU on gfx950


// The selection knobs (top-k, sink/recent pages, page cap, page size) are one-shot: the next phase 1 of an op that
// reads them takes them and restores the defaults, whether that call succeeds or not, so a value set for one call can
// never leak into a later, unrelated call on the same thread.  The launch-planning knobs (split, tail sharing, shared
// scoring) do not change what is computed beyond fp32 rounding and stay until changed.
struct SelectionKnobs { int64_t topk; int sink_recent, page_cap; int64_t page_size; };
SelectionKnobs take_selection_knobs() {
    const SelectionKnobs k{g_topk, g_sink_recent, g_page_cap, g_page_size};
    g_topk = 64;
    g_sink_recent = 0;
    g_page_cap = 0;
    g_page_size = 128;
    return k;
}

// One-shot operands of the two attention ops that the aclnn signatures have no slot for (taken before anything can
// fail, so a rejected call never leaves a device pointer armed for the next one).
struct AttentionExtras { float* lse; int64_t lse_n; SharedPrefix prefix; int antiq_mode; };
AttentionExtras take_attention_extras() {
    const AttentionExtras x{g_lse_out, g_lse_out_n, g_prefix, g_antiq_mode};
    g_antiq_mode = 0;
    g_lse_out = nullptr;
    g_lse_out_n = 0;
    g_prefix = SharedPrefix{nullptr, nullptr, 0, 0};
    return x;
}

int current_device_or_none() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return dev;
}

aclnnStatus fail(aclnnStatus code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define REQUIRE(cond, ...)                                            \
    do {                                                              \
        if (!(cond)) return fail(ACLNN_ERR_PARAM_INVALID, __VA_ARGS__); \
    } while (0)
#define REQUIRE_PTR(p, name)                                                         \
    do {                                                                             \
        if ((p) == nullptr) return fail(ACLNN_ERR_PARAM_NULLPTR, "%s must not be NULL", name); \
    } while (0)

int elem_size(aclDataType t) {
    switch (t) {
        case ACL_FLOAT: case ACL_INT32: case ACL_UINT32: return 4;
        case ACL_FLOAT16: case ACL_BF16: case ACL_INT16: case ACL_UINT16: return 2;
        case ACL_INT8: case ACL_UINT8: case ACL_BOOL: return 1;
        case ACL_INT64: case ACL_UINT64: case ACL_DOUBLE: return 8;
        default: return 0;
    }
}

int64_t numel(const aclTensor* t) {
    int64_t n = 1;
    for (int64_t d : t->dims) n *= d;
    return n;
}

bool is_contiguous(const aclTensor* t) {
    int64_t expect = 1;
    for (int i = (int)t->dims.size() - 1; i >= 0; --i) {
        if (t->dims[i] != 1 && t->strides[i] != expect) return false;
        expect *= t->dims[i];
    }
    return true;
}

void* data_ptr(const aclTensor* t) { return (char*)t->data + t->offset * elem_size(t->dtype); }

// Optional operands arrive as NULL (the binding maps numel()==0 tensors to NULL, add_custom.cpp:32-35);
// a descriptor with zero elements is treated the same way.
bool absent(const aclTensor* t) { return t == nullptr || numel(t) == 0 || t->data == nullptr; }

aclnnStatus check_tensor(const aclTensor* t, const char* name, size_t ndim, aclDataType dtype) {
    REQUIRE_PTR(t, name);
    REQUIRE(t->dims.size() == ndim, "%s: expected %zu dims, got %zu", name, ndim, t->dims.size());
    REQUIRE(t->dtype == dtype, "%s: unexpected dtype %d (expected %d)", name, (int)t->dtype, (int)dtype);
    REQUIRE(is_contiguous(t), "%s must be contiguous", name);
    REQUIRE(t->data != nullptr || numel(t) == 0, "%s has no device memory", name);
    return ACLNN_SUCCESS;
}
#define CHECK(expr)                           \
    do {                                      \
        aclnnStatus _s = (expr);              \
        if (_s != ACLNN_SUCCESS) return _s;   \
    } while (0)

int to_dtype(aclDataType t) {
    switch (t) {
        case ACL_FLOAT16: return spfa::kF16;
        case ACL_BF16: return spfa::kBF16;
        case ACL_INT8: return spfa::kI8;
        case ACL_FLOAT: return spfa::kF32;
        default: return -1;
    }
}

int round_up(int x, int m) { return (x + m - 1) / m * m; }

// q rows scored per centroid pass by the standalone scoring ops: the keys [rows][Cp] u32 and the eight waves' centroid
// tiles share a 100 KiB budget (8 rows at C = 2048, D = 128: one row per wave), at most 16 rows (one MFMA tile).  When a whole group fits, rows of several sequences share the pass, dealt
// evenly over the workgroups of a kv head.
// aux_bytes: what the op keeps in LDS besides keys and tiles (sort buffers, cluster bitmaps); the keys take what is left
// of the CU's LDS when that is less than the budget.
void plan_cent_rows(spfa::CentParams& p, size_t aux_bytes) {
#ifndef SPFA_CENT_BUDGET_KB
#define SPFA_CENT_BUDGET_KB 100
#endif
    p.score_waves = spfa::cent_score_waves(p.D);
    long long budget = SPFA_CENT_BUDGET_KB * 1024;
    if (budget > (long long)kMaxLds - (long long)aux_bytes) budget = (long long)kMaxLds - (long long)aux_bytes;
    int r = (int)((budget - spfa::cent_stage_bytes(p.D)) / ((long long)p.Cp * 4));
    if (r < 1) r = 1;
    if (r > 16) r = 16;
    if (p.G >= r) {                 // the group alone fills (or exceeds) a pass
        p.rows_per_pass = r;
        p.seq_per_group = 1;
    } else {
        p.seq_per_group = r / p.G;
        if (p.seq_per_group > p.B) p.seq_per_group = p.B > 0 ? p.B : 1;
        // a small launch is latency bound: rather one workgroup per CU re-reading centroids from L2 than a few
        // workgroups that each serve many sequences
        const int n_cu = spfa::device_cu_count();
        if (n_cu > 0 && p.Hkv > 0) {
            const int groups_wanted = (n_cu + p.Hkv - 1) / p.Hkv;
            const int cap = p.B / groups_wanted > 1 ? p.B / groups_wanted : 1;
            if (p.seq_per_group > cap) p.seq_per_group = cap;
        }
        const int groups = (p.B + p.seq_per_group - 1) / p.seq_per_group;   // same number of passes, equal sizes
        if (groups > 0) p.seq_per_group = (p.B + groups - 1) / groups;
        p.rows_per_pass = p.seq_per_group * p.G;
    }
}

aclnnStatus check_qcent(const aclTensor* query, const aclTensor* l1, int* Hq, int* Hkv, int* D, int* C) {
    REQUIRE_PTR(query, "query");
    REQUIRE_PTR(l1, "l1_cent");
    REQUIRE(query->dtype == ACL_FLOAT16 || query->dtype == ACL_BF16, "query must be float16 or bfloat16");
    REQUIRE(l1->dtype == query->dtype, "l1_cent dtype must match query (cent_select_def.cpp:12-21)");
    REQUIRE(l1->dims.size() == 3, "l1_cent must be [Hkv, C, D]");
    REQUIRE(is_contiguous(query) && is_contiguous(l1), "query / l1_cent must be contiguous");
    *Hkv = (int)l1->dims[0];
    *C = (int)l1->dims[1];
    *D = (int)l1->dims[2];
    REQUIRE(*Hkv > 0 && *C > 0, "l1_cent is empty");
    REQUIRE(*D == 32 || *D == 64 || *D == 128 || *D == 256 || *D == 512, "head dim %d: must be 32, 64, 128, 256 or 512", *D);
    REQUIRE(*C <= 16384, "cluster count %d exceeds 16384", *C);
    (void)Hq;
    return ACLNN_SUCCESS;
}

aclnnStatus check_page_tables(const aclTensor* ids, const aclTensor* table, const aclTensor* seq, int Hkv, int* B,
                              int* kv_page_len, int* max_page) {
    CHECK(check_tensor(ids, "block_ids", 2, ACL_INT32));
    CHECK(check_tensor(table, "block_table", 2, ACL_INT32));
    CHECK(check_tensor(seq, "seq_len", 1, ACL_INT32));
    REQUIRE(ids->dims[0] == Hkv, "block_ids has %lld rows, expected num kv heads %d", (long long)ids->dims[0], Hkv);
    *B = (int)seq->dims[0];   // batch comes from seq_len (cent_select_tiling.cpp:32,53)
    *kv_page_len = (int)ids->dims[1];
    *max_page = (int)table->dims[1];
    REQUIRE(*kv_page_len > 0, "block_ids has no pages");
    REQUIRE(table->dims[0] >= *B, "block_table has %lld rows < batch %d", (long long)table->dims[0], *B);
    return ACLNN_SUCCESS;
}

aclnnStatus hip_fail(hipError_t e, const char* what) {
    return fail(ACLNN_ERR_RUNTIME_ERROR, "%s: %s", what, hipGetErrorString(e));
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// descriptor shim
// ------------------------------------------------------------------------------------------------
extern "C" {

// library-owned counters (defined with the attention ops below)
static bool stream_is_capturing(hipStream_t stream);
static void* stream_slot(hipStream_t stream, uint64_t zero_bytes, bool* needs_zero);
static void stream_slot_failed(hipStream_t stream);

aclTensor* aclCreateTensor(const int64_t* viewDims, uint64_t viewDimsNum, aclDataType dataType, const int64_t* stride,
                           int64_t offset, aclFormat, const int64_t*, uint64_t, void* tensorData) {
    if (viewDimsNum > 0 && viewDims == nullptr) return nullptr;
    aclTensor* t = new aclTensor();
    t->dims.assign(viewDims, viewDims + viewDimsNum);
    if (stride) {
        t->strides.assign(stride, stride + viewDimsNum);
    } else {
        t->strides.resize(viewDimsNum);
        int64_t s = 1;
        for (int i = (int)viewDimsNum - 1; i >= 0; --i) { t->strides[i] = s; s *= viewDims[i]; }
    }
    t->offset = offset;
    t->dtype = dataType;
    t->data = tensorData;
    return t;
}
aclnnStatus aclDestroyTensor(const aclTensor* t) { delete t; return ACLNN_SUCCESS; }

aclTensorList* aclCreateTensorList(const aclTensor* const* value, uint64_t size) {
    aclTensorList* l = new aclTensorList();
    for (uint64_t i = 0; i < size; ++i) l->items.push_back(const_cast<aclTensor*>(value[i]));
    return l;
}
aclnnStatus aclDestroyTensorList(const aclTensorList* l) {
    if (l) for (aclTensor* t : l->items) delete t;
    delete l;
    return ACLNN_SUCCESS;
}
aclIntArray* aclCreateIntArray(const int64_t* value, uint64_t size) {
    aclIntArray* a = new aclIntArray();
    if (value) a->v.assign(value, value + size);
    return a;
}
aclnnStatus aclDestroyIntArray(const aclIntArray* a) { delete a; return ACLNN_SUCCESS; }

aclnnStatus aclSetAclOpExecutorRepeatable(aclOpExecutor* e) {
    if (!e) return fail(ACLNN_ERR_PARAM_NULLPTR, "executor is NULL");
    e->repeatable = true;
    return ACLNN_SUCCESS;
}
aclnnStatus aclDestroyAclOpExecutor(aclOpExecutor* e) { delete e; return ACLNN_SUCCESS; }
const char* aclGetRecentErrMsg(void) { return g_err.c_str(); }

int InitHugeMemThreadLocal(void*, bool) { return 0; }
void UnInitHugeMemThreadLocal(void*, bool) {}
void ReleaseHugeMem(void*, bool) {}

aclnnStatus spfaSetTopK(int64_t k) {
    if (k < 1) return fail(ACLNN_ERR_PARAM_INVALID, "top-k must be >= 1, got %lld", (long long)k);
    g_topk = k;
    return ACLNN_SUCCESS;
}
aclnnStatus spfaSetSinkRecent(int enable) { g_sink_recent = enable ? 1 : 0; return ACLNN_SUCCESS; }
aclnnStatus spfaSetPageCap(int enable) { g_page_cap = enable ? 1 : 0; return ACLNN_SUCCESS; }
aclnnStatus spfaSetMaxSplit(int64_t max_split) {
    if (max_split < 1 || max_split > 16) return fail(ACLNN_ERR_PARAM_INVALID, "max split %lld out of [1, 16]", (long long)max_split);
    g_max_split = (int)max_split;
    return ACLNN_SUCCESS;
}
aclnnStatus spfaSetTailSharing(int enable) { g_tail_sharing = enable < 0 ? 0 : (enable > 2 ? 2 : enable); return ACLNN_SUCCESS; }
aclnnStatus spfaSetSharedScoring(int enable) { g_shared_scoring = enable ? 1 : 0; return ACLNN_SUCCESS; }
aclnnStatus spfaSetWorkgroupGeometry(int geometry) {
    if (geometry < 0 || geometry > 2) return fail(ACLNN_ERR_PARAM_INVALID, "workgroup geometry %d: 0 (auto), 1 (wide) or 2 (narrow)", geometry);
    g_geometry = geometry;
    return ACLNN_SUCCESS;
}
aclnnStatus spfaSetBatchedSelection(int enable) { g_super_items = enable ? 1 : 0; return ACLNN_SUCCESS; }
aclnnStatus spfaSetKThroughLds(int enable) { g_k_through_lds = enable ? 1 : 0; return ACLNN_SUCCESS; }
aclnnStatus spfaSetLibraryCounters(int enable) { g_stream_slots = enable ? 1 : 0; return ACLNN_SUCCESS; }
aclnnStatus spfaSetPageSize(int64_t page_size) {
    if (page_size < 1 || page_size > 512) return fail(ACLNN_ERR_PARAM_INVALID, "page size %lld out of [1, 512]", (long long)page_size);
    g_page_size = page_size;
    return ACLNN_SUCCESS;
}
aclnnStatus spfaSetActualSeqLengthsDevice(const int64_t* dev, int64_t n) {
    g_act_seq_dev = (const long long*)dev;
    g_act_seq_dev_n = dev ? n : 0;
    return ACLNN_SUCCESS;
}
aclnnStatus spfaSetTokenPositions(int enable) { g_token_positions = enable ? 1 : 0; return ACLNN_SUCCESS; }
aclnnStatus spfaSetSharedPrefix(const void* keyPrefix, const void* valuePrefix, int64_t prefixTokens, int64_t actualPrefixLen) {
    if ((keyPrefix == nullptr) != (valuePrefix == nullptr))
        return fail(ACLNN_ERR_PARAM_INVALID, "key and value shared prefix must be given together");
    if (keyPrefix && (prefixTokens < 1 || actualPrefixLen < 0 || actualPrefixLen > prefixTokens))
        return fail(ACLNN_ERR_PARAM_INVALID, "shared prefix of %lld tokens with actual length %lld", (long long)prefixTokens,
                    (long long)actualPrefixLen);
    g_prefix = keyPrefix ? SharedPrefix{keyPrefix, valuePrefix, prefixTokens, actualPrefixLen} : SharedPrefix{nullptr, nullptr, 0, 0};
    return ACLNN_SUCCESS;
}
aclnnStatus spfaSetAntiquantMode(int64_t mode) {
    if (mode < 0 || mode > 3)
        return fail(ACLNN_ERR_PARAM_INVALID,
                    "antiquant mode %lld: 0 (per channel / per tensor), 1 (per token), 2 (per tensor and head) or 3 (per token "
                    "and head); the modes that keep the parameters in page-attention layout (4, 5) are not supported",
                    (long long)mode);
    g_antiq_mode = (int)mode;
    return ACLNN_SUCCESS;
}
aclnnStatus spfaSetSoftmaxLseOut(float* deviceLse, int64_t n) {
    g_lse_out = deviceLse;
    g_lse_out_n = deviceLse ? n : 0;
    return ACLNN_SUCCESS;
}
aclnnStatus spfaSetWorkspacePersistent(aclOpExecutor* e, int enable) {
    if (!e) return fail(ACLNN_ERR_PARAM_NULLPTR, "executor is NULL");
    if (enable && !e->repeatable) return fail(ACLNN_ERR_PARAM_INVALID, "only a repeatable executor can keep a workspace");
    e->persistent_ws = enable != 0;
    e->clean_ws = nullptr;
    return ACLNN_SUCCESS;
}
// Diagnostic only: device buffer of >= grid*16*16 uint64 that -DSPFA_STAMPS builds fill with cycle stamps.
__attribute__((visibility("default"))) aclnnStatus spfaSetDebugBuffer(void* dev) { g_debug_buf = dev; return ACLNN_SUCCESS; }
// The launch plan of an executor of the two attention ops as text (for logs and tests): which kernel form and geometry,
// how the work is dealt.  Returns the length written (without the terminator), or -1.
int spfaDescribeExecutor(const aclOpExecutor* e, char* buf, uint64_t n) {
    if (!e || !buf || n == 0 || (e->op != kOpAttention && e->op != kOpFusion)) return -1;
    const spfa::AttnParams& p = e->attn;
    const bool mfma = spfa::attention_uses_mfma(p.rows_per_pass, p.kv_dtype, p.D);
    const int w = snprintf(buf, n,
                           "form=%s geometry=%s rows_per_pass=%d scored_ahead=%d lists_ahead=%d super_items=%d split=%d "
                           "tail_sharing=%d resident_workgroups=%lld lds_bytes=%zu workspace_bytes=%llu zeroed_bytes=%llu "
                           "token_window=%d k_through_lds=%d",
                           mfma ? "mfma" : "valu", p.narrow ? "narrow" : "wide", p.rows_per_pass, p.pre_items, p.pre_lists,
                           p.sup_items, p.n_split, p.tail_on, p.grid_cap, p.lds_bytes, (unsigned long long)e->ws_bytes,
                           (unsigned long long)e->ws_zero_bytes, p.token_mode ? p.max_entries : 0, mfma ? p.k_lds : 0);
    return w < 0 ? -1 : (w < (int)n ? w : (int)n - 1);
}
uint64_t spfaKernelLaunches(void) { return spfa::launches_so_far(); }
const char* spfaVersion(void) { return "spfa-hip-gfx950 0.1"; }

// ------------------------------------------------------------------------------------------------
// ComputeCent
// ------------------------------------------------------------------------------------------------
aclnnStatus aclnnComputeCentGetWorkspaceSize(const aclTensor* query, const aclTensor* l1Cent, const aclTensor* indicesOut,
                                             uint64_t* workspaceSize, aclOpExecutor** executor) {
    REQUIRE_PTR(workspaceSize, "workspaceSize");
    REQUIRE_PTR(executor, "executor");
    int Hq = 0, Hkv = 0, D = 0, C = 0;
    CHECK(check_qcent(query, l1Cent, &Hq, &Hkv, &D, &C));
    REQUIRE(query->dims.size() == 3 && query->dims[2] == D, "query must be [B, Hq, %d]", D);
    CHECK(check_tensor(indicesOut, "indices", 3, ACL_INT32));
    const int B = (int)query->dims[0];
    Hq = (int)query->dims[1];
    REQUIRE(Hq % Hkv == 0, "num heads %d not divisible by num kv heads %d", Hq, Hkv);
    REQUIRE(indicesOut->dims[0] == B && indicesOut->dims[1] == Hq, "indices must be [B, Hq, k]");
    const int k = (int)indicesOut->dims[2];   // compute_cent_tiling.cpp:45
    REQUIRE(k >= 1 && k <= C && k <= 1024, "k = %d must be in [1, min(C = %d, 1024)]", k, C);
    aclOpExecutor* e = new aclOpExecutor();
    e->op = kOpComputeCent;
    spfa::CentParams& p = e->cent;
    p.q = data_ptr(query);
    p.cent = data_ptr(l1Cent);
    p.qk_dtype = to_dtype(query->dtype);
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.G = Hq / Hkv; p.D = D; p.C = C; p.k = k;
    p.Cp = round_up(C, 128);
    p.indices = (int32_t*)data_ptr(indicesOut);
    int npow2 = 64;
    while (npow2 < k) npow2 <<= 1;
    size_t aux = (size_t)spfa::kCentWaves * npow2 * 8;   // one sort buffer per wave
    plan_cent_rows(p, aux);
    // wave w sorts rows w, w + kCentWaves, ...: a pass of fewer rows than waves (many clusters: one 64 KiB key row at
    // C = 16384) needs that many buffers only
    if (p.rows_per_pass < spfa::kCentWaves) aux = (size_t)p.rows_per_pass * npow2 * 8;
    p.lds_bytes = (size_t)p.rows_per_pass * p.Cp * 4 + aux + (size_t)spfa::cent_stage_bytes(p.D);
    if (p.lds_bytes > kMaxLds) {
        delete e;
        return fail(ACLNN_ERR_PARAM_INVALID, "compute_cent with %d clusters, k = %d, head dim %d needs %zu bytes of LDS (> %zu)", C, k, D,
                    p.lds_bytes, kMaxLds);
    }
    *workspaceSize = 0;
    *executor = e;
    return ACLNN_SUCCESS;
}

static aclnnStatus finish(aclOpExecutor* e, hipError_t err, const char* what) {
    if (!e->repeatable) delete e;
    if (err != hipSuccess) return hip_fail(err, what);
    return ACLNN_SUCCESS;
}

aclnnStatus aclnnComputeCent(void*, uint64_t, aclOpExecutor* executor, aclrtStream stream) {
    REQUIRE_PTR(executor, "executor");
    REQUIRE(executor->op == kOpComputeCent, "executor was not created by aclnnComputeCentGetWorkspaceSize");
    return finish(executor, spfa::launch_compute_cent(executor->cent, (hipStream_t)stream), "compute_cent launch");
}

// ------------------------------------------------------------------------------------------------
// SelectPosition
// ------------------------------------------------------------------------------------------------
aclnnStatus aclnnSelectPositionGetWorkspaceSize(const aclTensor* blockIds, const aclTensor* blockTable,
                                                const aclTensor* seqLen, const aclTensor* indices,
                                                const aclTensor* pagePositionOut, const aclTensor* pagePositionLengthOut,
                                                const aclTensor* blockTableGatherOut, uint64_t* workspaceSize,
                                                aclOpExecutor** executor) {
    const SelectionKnobs knobs = take_selection_knobs();
    REQUIRE_PTR(workspaceSize, "workspaceSize");
    REQUIRE_PTR(executor, "executor");
    CHECK(check_tensor(indices, "indices", 3, ACL_INT32));
    REQUIRE_PTR(blockIds, "block_ids");
    const int Hkv = blockIds->dims.empty() ? 0 : (int)blockIds->dims[0];
    int B = 0, kv_page_len = 0, max_page = 0;
    CHECK(check_page_tables(blockIds, blockTable, seqLen, Hkv, &B, &kv_page_len, &max_page));
    const int Hq = (int)indices->dims[1], k = (int)indices->dims[2];
    REQUIRE(Hkv > 0 && Hq % Hkv == 0, "num heads %d not divisible by num kv heads %d", Hq, Hkv);
    REQUIRE(indices->dims[0] >= B, "indices batch %lld < seq_len batch %d", (long long)indices->dims[0], B);
    REQUIRE(k >= 1 && k <= 4096, "k = %d out of range", k);
    CHECK(check_tensor(pagePositionOut, "page_position", 3, ACL_INT32));
    CHECK(check_tensor(pagePositionLengthOut, "page_position_length", 3, ACL_INT32));
    CHECK(check_tensor(blockTableGatherOut, "block_table_gather", 3, ACL_INT32));
    REQUIRE(pagePositionOut->dims[0] >= B && pagePositionOut->dims[1] == Hq && pagePositionOut->dims[2] >= 1,
            "page_position must be [B, Hq, P]");
    REQUIRE(pagePositionLengthOut->dims[0] >= B && pagePositionLengthOut->dims[1] == Hq &&
                pagePositionLengthOut->dims[2] == spfa::kLenPad, "page_position_length must be [B, Hq, 8]");
    REQUIRE(blockTableGatherOut->dims[0] >= B && blockTableGatherOut->dims[1] == Hq &&
                blockTableGatherOut->dims[2] == max_page, "block_table_gather must be [B, Hq, %d]", max_page);
    aclOpExecutor* e = new aclOpExecutor();
    e->op = kOpSelectPosition;
    spfa::SelectParams& p = e->sel;
    p.block_ids = (const int32_t*)data_ptr(blockIds);
    p.block_table = (const int32_t*)data_ptr(blockTable);
    p.seq_len = (const int32_t*)data_ptr(seqLen);
    p.indices = (const int32_t*)data_ptr(indices);
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.G = Hq / Hkv; p.k = k;
    p.kv_page_len = kv_page_len; p.max_page = max_page; p.block_size = (int)knobs.page_size;
    p.P = (int)pagePositionOut->dims[2];
    p.page_position = (int32_t*)data_ptr(pagePositionOut);
    p.page_position_length = (int32_t*)data_ptr(pagePositionLengthOut);
    p.block_table_gather = (int32_t*)data_ptr(blockTableGatherOut);
    *workspaceSize = 0;
    *executor = e;
    return ACLNN_SUCCESS;
}

aclnnStatus aclnnSelectPosition(void*, uint64_t, aclOpExecutor* executor, aclrtStream stream) {
    REQUIRE_PTR(executor, "executor");
    REQUIRE(executor->op == kOpSelectPosition, "executor was not created by aclnnSelectPositionGetWorkspaceSize");
    return finish(executor, spfa::launch_select_position(executor->sel, (hipStream_t)stream), "select_position launch");
}

// ------------------------------------------------------------------------------------------------
// CentSelect
// ------------------------------------------------------------------------------------------------
aclnnStatus aclnnCentSelectGetWorkspaceSize(const aclTensor* query, const aclTensor* l1Cent, const aclTensor* blockIds,
                                            const aclTensor* blockTable, const aclTensor* seqLen, const aclTensor*,
                                            const aclTensor* pagePositionOut, const aclTensor* pagePositionLengthOut,
                                            const aclTensor* maxPagePositionLengthOut, uint64_t* workspaceSize,
                                            aclOpExecutor** executor) {
    const SelectionKnobs knobs = take_selection_knobs();
    REQUIRE_PTR(workspaceSize, "workspaceSize");
    REQUIRE_PTR(executor, "executor");
    int Hq = 0, Hkv = 0, D = 0, C = 0;
    CHECK(check_qcent(query, l1Cent, &Hq, &Hkv, &D, &C));
    REQUIRE(query->dims.size() == 3 && query->dims[2] == D, "query must be [B, Hq, %d]", D);
    Hq = (int)query->dims[1];
    REQUIRE(Hq % Hkv == 0, "num heads %d not divisible by num kv heads %d", Hq, Hkv);
    int B = 0, kv_page_len = 0, max_page = 0;
    CHECK(check_page_tables(blockIds, blockTable, seqLen, Hkv, &B, &kv_page_len, &max_page));
    REQUIRE(query->dims[0] >= B, "query batch %lld < seq_len batch %d", (long long)query->dims[0], B);
    const int k = (int)knobs.topk;
    REQUIRE(k <= C, "top-k %d exceeds the cluster count %d", k, C);
    CHECK(check_tensor(pagePositionOut, "page_position", 3, ACL_INT32));
    CHECK(check_tensor(pagePositionLengthOut, "page_position_length", 3, ACL_INT32));
    CHECK(check_tensor(maxPagePositionLengthOut, "max_page_position_length", 2, ACL_INT64));
    REQUIRE(pagePositionOut->dims[0] >= B && pagePositionOut->dims[1] == Hq && pagePositionOut->dims[2] >= 1,
            "page_position must be [B, Hq, P]");
    REQUIRE(pagePositionLengthOut->dims[0] >= B && pagePositionLengthOut->dims[1] == Hq &&
                pagePositionLengthOut->dims[2] == spfa::kLenPad, "page_position_length must be [B, Hq, 8]");
    REQUIRE(maxPagePositionLengthOut->dims[0] >= B && maxPagePositionLengthOut->dims[1] == spfa::kLenPad,
            "max_page_position_length must be [B, 8]");
    aclOpExecutor* e = new aclOpExecutor();
    e->op = kOpCentSelect;
    spfa::CentParams& p = e->cent;
    p.q = data_ptr(query);
    p.cent = data_ptr(l1Cent);
    p.qk_dtype = to_dtype(query->dtype);
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.G = Hq / Hkv; p.D = D; p.C = C; p.k = k;
    p.Cp = round_up(C, 128);
    size_t aux = (size_t)spfa::kCentWaves * (p.Cp / 32) * 4;   // one cluster bitmap per wave
    plan_cent_rows(p, aux);
    if (p.rows_per_pass < spfa::kCentWaves) aux = (size_t)p.rows_per_pass * (p.Cp / 32) * 4;   // (a wave per row of the pass)
    p.block_ids = (const int32_t*)data_ptr(blockIds);
    p.block_table = (const int32_t*)data_ptr(blockTable);
    p.seq_len = (const int32_t*)data_ptr(seqLen);
    p.kv_page_len = kv_page_len; p.max_page = max_page; p.block_size = (int)knobs.page_size;
    p.P = (int)pagePositionOut->dims[2];
    p.sink_recent = knobs.sink_recent; p.cap_pages = knobs.page_cap;
    p.page_position = (int32_t*)data_ptr(pagePositionOut);
    p.page_position_length = (int32_t*)data_ptr(pagePositionLengthOut);
    p.max_page_position_length = (unsigned long long*)data_ptr(maxPagePositionLengthOut);
    p.lds_bytes = (size_t)p.rows_per_pass * p.Cp * 4 + aux + (size_t)spfa::cent_stage_bytes(p.D);
    if (p.lds_bytes > kMaxLds) {
        delete e;
        return fail(ACLNN_ERR_PARAM_INVALID, "cent_select with %d clusters, head dim %d needs %zu bytes of LDS (> %zu)", C, D,
                    p.lds_bytes, kMaxLds);
    }
    *workspaceSize = 0;
    *executor = e;
    return ACLNN_SUCCESS;
}

aclnnStatus aclnnCentSelect(void*, uint64_t, aclOpExecutor* executor, aclrtStream stream) {
    REQUIRE_PTR(executor, "executor");
    REQUIRE(executor->op == kOpCentSelect, "executor was not created by aclnnCentSelectGetWorkspaceSize");
    spfa::CentParams p = executor->cent;
    hipError_t err = hipSuccess;
    // The per-batch maximum over a sequence's Hq rows, which different workgroups select.  With a slot of library-owned
    // counters for this stream (see stream_slot): a running maximum and an arrival count per sequence, the row that
    // completes a sequence stores the result and clears the two words again — one kernel.  Otherwise (launch being
    // captured, batch beyond the slot, no slot left): atomicMax straight into the output, zeroed by a kernel in front.
    auto up256 = [](uint64_t x) { return (x + 255) & ~255ull; };
    const uint64_t slot_bytes = up256((uint64_t)p.B * 4) + (uint64_t)p.B * 8;
    bool needs_zero = false, in_slot = false;
    if (p.B > 0 && !stream_is_capturing((hipStream_t)stream)) {
        if (void* slot = stream_slot((hipStream_t)stream, slot_bytes, &needs_zero)) {
            in_slot = true;
            if (needs_zero) err = spfa::launch_zero(slot, spfa::kScratchSlotBytes, nullptr, 0, (hipStream_t)stream);
            p.batch_arrive = (unsigned*)slot;
            p.batch_max = (unsigned long long*)((char*)slot + up256((uint64_t)p.B * 4));
        }
    }
    if (!in_slot && p.B > 0)
        err = spfa::launch_zero(p.max_page_position_length, (size_t)p.B * spfa::kLenPad * 8, nullptr, 0, (hipStream_t)stream);
    if (err == hipSuccess) err = spfa::launch_cent_select(p, (hipStream_t)stream);
    if (err != hipSuccess && in_slot) stream_slot_failed((hipStream_t)stream);
    return finish(executor, err, "cent_select launch");
}

// ------------------------------------------------------------------------------------------------
// SparsePagedAttention / SparsePagedFusionAttention
// ------------------------------------------------------------------------------------------------
static aclnnStatus plan_attention(spfa::AttnParams& p, const aclTensor* query, const aclTensorList* key,
                                  const aclTensorList* value, const aclTensor* antiquantScale,
                                  const aclTensor* antiquantOffset, const aclTensor* blocktable, int64_t numHeads,
                                  double scaleValue, const char* inputLayout, int64_t numKeyValueHeads,
                                  int64_t blockSize, const aclTensor* attentionOut, int batch_limit, int antiq_mode) {
    REQUIRE_PTR(query, "query");
    REQUIRE(query->dtype == ACL_FLOAT16 || query->dtype == ACL_BF16, "query must be float16 or bfloat16");
    REQUIRE(is_contiguous(query), "query must be contiguous");
    REQUIRE(numHeads > 0, "numHeads must be positive");
    const int Hq = (int)numHeads;
    const int Hkv = numKeyValueHeads > 0 ? (int)numKeyValueHeads : Hq;
    REQUIRE(Hq % Hkv == 0, "numHeads %d not divisible by numKeyValueHeads %d", Hq, Hkv);
    REQUIRE(Hq / Hkv <= 64, "group size %d exceeds 64 (…_tiling_check.cc:171)", Hq / Hkv);
    const std::string layout = inputLayout ? inputLayout : "BNSD";
    REQUIRE(layout == "BSH" || layout == "BNSD" || layout == "BSND",
            "Only support input_layout(BSH, BNSD, BSND), actually is %s (…_tiling.cc:226)", layout.c_str());
    // decode only: S = 1, so every accepted layout is [B, Hq, D] in memory
    int B = 0, D = 0;
    const auto& qd = query->dims;
    if (qd.size() == 3 && layout == "BSH") {            // [B, 1, Hq*D]
        REQUIRE(qd[1] == 1 && qd[2] % Hq == 0, "BSH query must be [B, 1, numHeads*D]");
        B = (int)qd[0]; D = (int)(qd[2] / Hq);
    } else if (qd.size() == 3) {                        // [B, Hq, D] (…attention_tiling.cc:207-225)
        REQUIRE(qd[1] == Hq, "query must be [B, numHeads = %d, D]", Hq);
        B = (int)qd[0]; D = (int)qd[2];
    } else if (qd.size() == 4 && layout == "BSND") {    // [B, 1, Hq, D]
        REQUIRE(qd[1] == 1 && qd[2] == Hq, "BSND query must be [B, 1, numHeads, D]");
        B = (int)qd[0]; D = (int)qd[3];
    } else if (qd.size() == 4) {                        // BNSD [B, Hq, 1, D]
        REQUIRE(qd[1] == Hq && qd[2] == 1, "BNSD query must be [B, numHeads, 1, D] (decode: one query token)");
        B = (int)qd[0]; D = (int)qd[3];
    } else {
        return fail(ACLNN_ERR_PARAM_INVALID, "query must have 3 or 4 dims");
    }
    REQUIRE(B <= 65536, "batch size %d cannot be greater than 65536 (…_tiling_check.cc:215)", B);
    REQUIRE(D > 0 && D <= 512, "head dim %d out of range (…_tiling_check.cc:219)", D);
    REQUIRE(Hkv <= 256, "numKeyValueHeads %d exceeds 256 (…_tiling_check.cc:222)", Hkv);

    REQUIRE(key && key->items.size() == 1 && value && value->items.size() == 1,
            "key / value must be one-tensor lists holding the paged cache");
    const aclTensor* kc = key->items[0];
    const aclTensor* vc = value->items[0];
    REQUIRE(kc->dtype == vc->dtype && kc->dims == vc->dims, "key and value cache must agree in dtype and shape");
    REQUIRE(is_contiguous(kc) && is_contiguous(vc), "key / value cache must be contiguous");
    REQUIRE(kc->data && vc->data, "key / value cache has no device memory");
    const bool quant = kc->dtype == ACL_INT8;
    REQUIRE(quant || kc->dtype == query->dtype,
            "KV cache dtype must equal the query dtype or be int8 (…_tiling_check.cc:178-182)");
    const int es = quant ? 1 : 2;
    REQUIRE(blockSize > 0 && blockSize <= 512, "blockSize %lld out of (0, 512] (…_tiling_check.cc:28-48)", (long long)blockSize);
    REQUIRE(blockSize % (quant ? 32 : 16) == 0, "blockSize %lld must be a multiple of %d (…_tiling_check.cc:188-191)",
            (long long)blockSize, quant ? 32 : 16);
    const int bs = (int)blockSize;
    const auto& kd = kc->dims;
    REQUIRE(kd.size() != 4 || layout == "BNSD",
            "with a 4-dim paged KV cache the query layout must be BNSD (…_tiling.cc:341-343)");
    bool bsh;
    if (kd.size() == 3) {
        REQUIRE(kd[1] == bs && kd[2] == (int64_t)Hkv * D, "KV cache must be [blocks, blockSize, Hkv*D]");
        bsh = true;
    } else if (kd.size() == 4 && kd[1] == bs && kd[2] == Hkv && kd[3] == D) {
        bsh = true;    // [blocks, bs, Hkv, D]: the vLLM layout the fused op forces (…_tiling.cc:338)
    } else if (kd.size() == 4 && kd[1] == Hkv && kd[2] == bs && kd[3] == D) {
        bsh = false;   // [blocks, Hkv, bs, D]
    } else {
        return fail(ACLNN_ERR_PARAM_INVALID, "KV cache must be [blocks,bs,Hkv*D], [blocks,bs,Hkv,D] or [blocks,Hkv,bs,D]");
    }
    const int lpt = D * es / 16;
    REQUIRE((D * es) % 16 == 0 && (lpt == 4 || lpt == 8 || lpt == 16 || lpt == 32 || (lpt == 64 && !quant)),
            "head dim %d with %d-byte KV elements is not supported (row bytes must be 64, 128, 256, 512 or 1024)", D, es);
    REQUIRE(kd[0] > 0 && kd[0] <= 0x7fffffff, "KV cache has no blocks");

    REQUIRE(antiq_mode == 0 || quant, "antiquant modes other than 0 need an int8 KV cache");
    if (quant && (antiq_mode == 1 || antiq_mode == 3)) {
        // per token (…_tiling.cc:733-765, …Us2.h:2039-2043): float32 [2, B, S], key parameters first; per token and head
        // (mode 3, …_tiling_check.cc:466-487, …Us2.h:2203-2206): [2, B, Hkv, S].  S is checked once the table width is known
        const size_t nd = antiq_mode == 3 ? 4 : 3;
        REQUIRE(!absent(antiquantScale), "int8 KV cache needs antiquantScale");
        REQUIRE(antiquantScale->dtype == ACL_FLOAT && antiquantScale->dims.size() == nd && antiquantScale->dims[0] == 2 &&
                    is_contiguous(antiquantScale) && antiquantScale->data,
                "per-token antiquantScale must be a contiguous float32 %s tensor (…_tiling.cc:733-765)",
                antiq_mode == 3 ? "[2, B, Hkv, S]" : "[2, B, S]");
        REQUIRE(antiq_mode != 3 || antiquantScale->dims[2] == Hkv,
                "per-token-head antiquantScale holds %lld heads, the op %d kv heads (…_tiling_check.cc:476)",
                (long long)antiquantScale->dims[2], Hkv);
        if (!absent(antiquantOffset))
            REQUIRE(antiquantOffset->dtype == ACL_FLOAT && antiquantOffset->dims == antiquantScale->dims &&
                        is_contiguous(antiquantOffset) && antiquantOffset->data,
                    "per-token antiquantOffset must have antiquantScale's shape and dtype");
        p.antiq_per_token = 1;
        p.antiq_s = antiquantScale->dims[nd - 1];
        p.antiq_tok_heads = antiq_mode == 3 ? Hkv : 0;
    } else if (quant && antiq_mode == 2) {
        // per tensor and head (…_tiling_check.cc:488-497: one value per kv head; here key values first, then value values)
        REQUIRE(!absent(antiquantScale), "int8 KV cache needs antiquantScale");
        REQUIRE(antiquantScale->dtype == query->dtype && antiquantScale->dims.size() == 2 && antiquantScale->dims[0] == 2 &&
                    antiquantScale->dims[1] == Hkv && is_contiguous(antiquantScale),
                "per-tensor-head antiquantScale must be [2, Hkv = %d] in the query dtype (…_tiling_check.cc:488-497)", Hkv);
        if (!absent(antiquantOffset))
            REQUIRE(antiquantOffset->dtype == query->dtype && antiquantOffset->dims == antiquantScale->dims &&
                        is_contiguous(antiquantOffset),
                    "antiquantOffset must have antiquantScale's shape and the query dtype");
        p.antiq_per_tensor = 2;
    } else if (quant) {
        REQUIRE(!absent(antiquantScale), "int8 KV cache needs antiquantScale");
        // [2]: one key and one value parameter for the whole tensor (…_tiling.cc:785-793; the value parameter is
        // element 1, …Us2.h:2036-2038), else per channel
        const bool per_tensor = antiquantScale->dims.size() == 1 && antiquantScale->dims[0] == 2;
        const long long want = per_tensor ? 2 : 2ll * Hkv * D;
        REQUIRE(antiquantScale->dtype == query->dtype && numel(antiquantScale) == want && is_contiguous(antiquantScale),
                "antiquantScale must be [2, Hkv, D] or [2] in the query dtype (…_tiling_check.cc:497-525)");
        if (!absent(antiquantOffset))
            REQUIRE(antiquantOffset->dtype == query->dtype && numel(antiquantOffset) == want &&
                        antiquantOffset->dims.size() == antiquantScale->dims.size() && is_contiguous(antiquantOffset),
                    "antiquantOffset must have antiquantScale's shape and the query dtype");
        p.antiq_per_tensor = per_tensor ? 1 : 0;
    } else {
        REQUIRE(absent(antiquantScale) && absent(antiquantOffset),
                "antiquant parameters given but the KV cache is not int8 (…_tiling.cc:862-866)");
    }
    CHECK(check_tensor(blocktable, "blocktable", 2, ACL_INT32));
    REQUIRE_PTR(attentionOut, "attentionOut");
    REQUIRE(attentionOut->dtype == query->dtype || attentionOut->dtype == ACL_FLOAT || attentionOut->dtype == ACL_INT8,
            "attentionOut dtype must equal the query dtype (or float32, or int8 with quantScale2)");
    REQUIRE(numel(attentionOut) == numel(query) && is_contiguous(attentionOut), "attentionOut must have query's shape");
    if (batch_limit >= 0) {
        REQUIRE(B >= batch_limit, "query batch %d < total_seq_len batch %d", B, batch_limit);
        B = batch_limit;
    }
    REQUIRE(blocktable->dims[0] >= B, "blocktable has %lld rows < batch %d", (long long)blocktable->dims[0], B);
    if (p.antiq_per_token) {
        REQUIRE(antiquantScale->dims[1] == B, "per-token antiquantScale holds %lld batches, the op %d (…_tiling.cc:740)",
                (long long)antiquantScale->dims[1], B);
        REQUIRE(p.antiq_s >= blocktable->dims[1] * blockSize,
                "per-token antiquant S %lld is less than sMax %lld (blocktable columns * blockSize, …_tiling.cc:745)",
                p.antiq_s, (long long)(blocktable->dims[1] * blockSize));
    }

    p.q = data_ptr(query);
    p.out = data_ptr(attentionOut);
    p.q_dtype = to_dtype(query->dtype);
    p.out_dtype = to_dtype(attentionOut->dtype);
    p.kv_dtype = to_dtype(kc->dtype);
    p.kcache = data_ptr(kc);
    p.vcache = data_ptr(vc);
    if (bsh) {
        p.blk_stride = (long long)bs * Hkv * D * es; p.tok_stride = (long long)Hkv * D * es; p.head_stride = (long long)D * es;
    } else {
        p.blk_stride = (long long)Hkv * bs * D * es; p.head_stride = (long long)bs * D * es; p.tok_stride = (long long)D * es;
    }
    p.n_blocks = (int)kd[0];
    p.block_table = (const int32_t*)data_ptr(blocktable);
    p.max_page = (int)blocktable->dims[1];
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.G = Hq / Hkv; p.D = D; p.block_size = bs;
    p.scale = (float)scaleValue;
    p.antiq_scale = quant ? data_ptr(antiquantScale) : nullptr;
    p.antiq_offset = (quant && !absent(antiquantOffset)) ? data_ptr(antiquantOffset) : nullptr;
    p.debug = (unsigned long long*)g_debug_buf;
    return ACLNN_SUCCESS;
}

// Optional operands of the attention core that ride on the same kernels: pseShift / attenMask over the columns of the
// gathered sequence (…_tiling_check.cc:737-844) and the int8 output quantisation (…_tiling.cc:569-650,
// …_tiling_check.cc:412-464).  Called once the batch and the table width are known.
static aclnnStatus plan_extras(spfa::AttnParams& p, const aclTensor* query, const aclTensor* pseShift,
                               const aclTensor* attenMask, const aclTensor* quantScale2, const aclTensor* quantOffset2,
                               const aclTensor* attentionOut) {
    const long long s_max = (long long)p.max_page * p.block_size;   // sMax of the paged cache
    if (!absent(pseShift)) {
        REQUIRE(pseShift->dtype == query->dtype, "pseShift dtype must equal the query dtype (…_tiling_check.cc:757)");
        REQUIRE(pseShift->dims.size() == 4 && is_contiguous(pseShift) && pseShift->data,
                "pseShift must be a contiguous [B | 1, numHeads, 1, S] tensor");
        const auto& d = pseShift->dims;
        REQUIRE((d[0] == 1 || d[0] == p.B) && d[1] == p.Hq && d[2] == 1,
                "pseShift is (%lld, %lld, %lld, %lld), which does not match (B, N, 1, S) or (1, N, 1, S)", (long long)d[0],
                (long long)d[1], (long long)d[2], (long long)d[3]);
        REQUIRE(d[3] >= s_max, "pseShift S %lld is less than sMax %lld (blocktable columns * blockSize)", (long long)d[3],
                s_max);
        REQUIRE(p.D % 16 == 0, "with pseShift the head dim must be 16 aligned");
        p.pse = data_ptr(pseShift);
        p.pse_b = (int)d[0];
        p.pse_s = d[3];
    }
    if (!absent(attenMask)) {
        REQUIRE(attenMask->dtype == ACL_BOOL || attenMask->dtype == ACL_INT8 || attenMask->dtype == ACL_UINT8,
                "attenMask must be bool, int8 or uint8 (…_tiling_check.cc:826-831)");
        REQUIRE(attenMask->dims.size() >= 2 && is_contiguous(attenMask) && attenMask->data,
                "attenMask must be a contiguous [B, ..., S] tensor");
        const long long s = attenMask->dims.back();
        REQUIRE(attenMask->dims[0] == p.B && numel(attenMask) == (long long)p.B * s,
                "attenMask must hold one row of S flags per batch (batch %d)", p.B);
        REQUIRE(s >= s_max, "attenMask S %lld is less than sMax %lld (blocktable columns * blockSize)", s, s_max);
        p.amask = (const unsigned char*)data_ptr(attenMask);
        p.amask_s = s;
    }
    if (attentionOut->dtype == ACL_INT8) {
        REQUIRE(!absent(quantScale2), "output data type is int8, but quantScale2 is absent (…_tiling.cc:572)");
        REQUIRE(quantScale2->dtype == ACL_FLOAT || quantScale2->dtype == ACL_BF16,
                "quantScale2 must be float32 or bfloat16 (…_tiling.cc:577)");
        REQUIRE(quantScale2->dtype != ACL_BF16 || query->dtype == ACL_BF16,
                "bfloat16 quantScale2 needs a bfloat16 query (…_tiling.cc:587)");
        const long long per_chn = (long long)p.Hq * p.D;
        const long long n = numel(quantScale2);
        REQUIRE((n == 1 || n == per_chn) && is_contiguous(quantScale2) && quantScale2->data,
                "quantScale2 must hold 1 or numHeads * D = %lld values (…_tiling_check.cc:412-464)", per_chn);
        p.quant_scale2 = data_ptr(quantScale2);
        p.quant2_dtype = to_dtype(quantScale2->dtype);
        p.quant2_per_channel = n > 1 ? 1 : 0;
        if (!absent(quantOffset2)) {
            REQUIRE(quantOffset2->dtype == quantScale2->dtype,
                    "quantScale2 and quantOffset2 must have the same dtype (…_tiling.cc:582-586)");
            REQUIRE(numel(quantOffset2) == n && is_contiguous(quantOffset2) && quantOffset2->data,
                    "quantOffset2 must have quantScale2's shape");
            p.quant_offset2 = data_ptr(quantOffset2);
        }
    } else {
        REQUIRE(absent(quantScale2), "output data type is not int8, but quantScale2 exists (…_tiling.cc:598)");
        REQUIRE(absent(quantOffset2), "output data type is not int8, but quantOffset2 exists (…_tiling.cc:601)");
    }
    return ACLNN_SUCCESS;
}

// The operands that arrive through one-shot setters instead of the aclnn argument list.
static aclnnStatus plan_oneshot_extras(spfa::AttnParams& p, const AttentionExtras& x) {
    if (x.lse) {
        REQUIRE(x.lse_n >= (int64_t)p.B * p.Hq, "softmax LSE output has %lld values < batch %d * numHeads %d",
                (long long)x.lse_n, p.B, p.Hq);
        REQUIRE(((uintptr_t)x.lse & 3) == 0, "softmax LSE output must be 4-byte aligned device memory");
        p.lse = x.lse;
    }
    REQUIRE(!(p.antiq_per_token && p.token_mode), "per-token antiquant parameters are not supported with token-level positions");
    if (x.prefix.k && x.prefix.len > 0) {
        // …_tiling.cc: the prefix tensors have the KV dtype and layout with batch 1; here they are addressed with the
        // cache's token stride, so only the distance between heads differs between the two layouts
        REQUIRE(!p.token_mode, "a shared prefix is not supported with token-level positions");
        REQUIRE(x.prefix.len <= 0x7fffffff / 2, "shared prefix of %lld tokens is too long", (long long)x.prefix.len);
        REQUIRE((((uintptr_t)x.prefix.k | (uintptr_t)x.prefix.v) & 15) == 0, "shared prefix tensors must be 16-byte aligned");
        const long long es = p.kv_dtype == spfa::kI8 ? 1 : 2;
        const bool bsh = p.Hkv == 1 || p.head_stride == (long long)p.D * es;
        p.prefix_k = x.prefix.k;
        p.prefix_v = x.prefix.v;
        p.prefix_head_stride = bsh ? (long long)p.D * es : x.prefix.tokens * p.D * es;
        p.prefix_len = (int)x.prefix.len;
        p.prefix_pages = (p.prefix_len + p.block_size - 1) / p.block_size;
        REQUIRE((long long)p.max_entries + p.prefix_pages <= 4095, "page list of %d entries plus %d prefix runs exceeds 4095",
                p.max_entries, p.prefix_pages);
        p.max_entries += p.prefix_pages;
        // pseShift / attenMask columns start with the prefix tokens (…Us2.h:2263-2266, 2283-2285)
        const long long s_need = (long long)p.max_page * p.block_size + p.prefix_len;
        REQUIRE(!p.pse || p.pse_s >= s_need, "pseShift S %lld is less than prefix + sMax = %lld", p.pse_s, s_need);
        REQUIRE(!p.amask || p.amask_s >= s_need, "attenMask S %lld is less than prefix + sMax = %lld", p.amask_s, s_need);
        REQUIRE(!p.antiq_per_token || p.antiq_s >= s_need, "per-token antiquant S %lld is less than prefix + sMax = %lld",
                p.antiq_s, s_need);
    }
    return ACLNN_SUCCESS;
}

constexpr int kTokenWindow = 4096;       // most entries of a token-level list held in LDS at a time (per pipeline slot)
constexpr uint64_t kCounterBytes = 256;   // 8 work counters at the start of the workspace, zeroed per launch

static size_t carve_lds(spfa::AttnParams& p, size_t nq) {
    auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    size_t off = 0;
    // rows whose score keys are held at once; rows whose page bitmaps are (super-items: the one being streamed and the
    // one being prepared)
    const size_t sel_rows = p.sup_items > 1 ? (size_t)p.sup_items : nq * (size_t)(p.pre_items > 1 ? p.pre_items : 1);
    const size_t bit_rows = p.sup_items > 1 ? 2 * (size_t)p.sup_items : sel_rows;
    const bool team = spfa::attention_team(p);   // a team of four selector waves: a cluster bitmap each
    p.off_keys = (int)off;
    if (p.mode == 1) off += align16(sel_rows * p.Cp * 4);                  // score keys (selector wave)
    p.off_cbits = (int)off;
    const size_t max_waves = 16;   // upper bound of the waves of a workgroup (streamers + selectors)
    if (p.mode == 1)   // top-k cluster bitmaps (one per wave that selects a row of the first item) + 16 row lengths
        off += align16((sel_rows < max_waves ? (team && sel_rows < 4 ? 4 : sel_rows) : max_waves) * (size_t)(p.Cp / 32) * 4) + 64;
    p.off_pbits = (int)off;
    if (p.mode == 1) off += align16(bit_rows * ((p.max_page + 63) / 64) * 8);   // per-head page bitmaps
    p.off_list = (int)off;
    off += align16((size_t)2 * p.max_entries * 8);                         // double-buffered page lists
    p.off_comb = (int)off;
    const bool mfma = spfa::attention_uses_mfma((int)nq, p.kv_dtype, p.D);
    const size_t sw = (size_t)spfa::attention_stream_waves(mfma, p.kv_dtype, p.narrow != 0);
    off += align16((size_t)2 * sw * nq * (p.D + 2) * 4);                   // streamer states of two items
    p.off_qst = (int)off;
    if (!mfma) off += align16((size_t)2 * nq * p.D * 4);                   // pre-scaled q rows (VALU streamer)
    p.off_carry = (int)off;
    if (p.token_mode && p.max_entries < round_up(p.P, p.block_size))       // the list is streamed in windows
        off += align16((size_t)2 * nq * (p.D + 2) * 4);
    p.off_hdr = (int)off;
    off += 2 * 48 + 16 + 64 + 144;                                         // two item headers + 4 workgroup words + 16 pipeline words + chunk table
    p.off_vst = (int)off;
    p.vst_stride = spfa::mfma_tile_area(p.D);
    if (mfma) off += sw * (size_t)p.vst_stride;                           // V (and K) tile images of the streamer waves
    p.off_stage = (int)off;
    if (p.mode == 1) off += (size_t)spfa::attention_stage_bytes(mfma, p.kv_dtype, p.D, p.narrow != 0, team);   // centroid tiles of the scoring waves
    p.score_waves = spfa::attention_score_waves(mfma, p.kv_dtype, p.D, p.narrow != 0, team);
    return off;
}

static aclnnStatus plan_lds(spfa::AttnParams& p) {
    const bool per_head = (p.mode == 0 && p.pos_rows == p.Hq && p.G > 1);
    // pseShift / attenMask / per-token antiquant parameters index the columns of ONE head's list: every q head runs
    // as its own pass
    // (token-level lists: a list per kv head is streamed once for the whole group; a list per q head is a pass of its own)
    int nq = (per_head || p.pse || p.amask || p.antiq_per_token) ? 1 : spfa::attention_rows_per_pass(p.G, p.kv_dtype, p.D);
    // Token-level lists: the whole list when it has at most kTokenWindow entries, else windows of that size (a window
    // boundary costs every streamer wave a drained load pipeline and a reload of its q rows: measured 2.5 % per boundary
    // on 100 us items); smaller windows before fewer q heads per pass where the areas of a wide pass leave less room.
    // The matrix-core streamer takes K through LDS for head dims up to 128 (the K image of a tile lies in the bytes of its
    // V image: no LDS of its own).
    p.k_lds = (g_k_through_lds && p.D <= 128) ? 1 : 0;
    size_t need = carve_lds(p, (size_t)nq);
    if (p.token_mode) {
        const int whole = round_up(p.P, p.block_size);
        const char* forced = getenv("SPFA_TOKEN_WINDOW");   // tests and experiments: the largest window
        const int w_max = forced && atoi(forced) > 0 ? atoi(forced) : kTokenWindow;
        int w = std::min(whole, std::max(p.block_size, w_max / p.block_size * p.block_size));
        for (;;) {
            p.max_entries = w;
            need = carve_lds(p, (size_t)nq);
            if (need <= kMaxLds || w <= std::max(p.block_size, 512)) break;
            w = std::max(p.block_size, w / 2 / p.block_size * p.block_size);
        }
    }
    // a large group with many clusters: fewer q heads per pass (more passes over the pages) until the keys fit
    while (need > kMaxLds && nq > 2 && spfa::attention_uses_mfma(nq, p.kv_dtype, p.D)) {
        nq = (nq + 1) / 2;
        need = carve_lds(p, (size_t)nq);
    }
    // ... and as the last resort every q head as a pass of its own (one key row: 64 KiB at the largest cluster count)
    if (need > kMaxLds && nq > 1) {
        nq = 1;
        need = carve_lds(p, 1);
    }
    p.rows_per_pass = nq;
    p.lds_bytes = need;
    p.pre_items = 0;
    REQUIRE(p.lds_bytes <= kMaxLds, "page list of %d entries / %d clusters needs %zu bytes of LDS (> %zu)", p.max_entries,
            p.C, p.lds_bytes, kMaxLds);
    return ACLNN_SUCCESS;
}

// Small batches leave most of the chip idle with one workgroup per item: deal each item's page list to several
// workgroups (the kernel merges their states in the same launch).  Decided here because the workspace holds the
// states; needs the CU count of the current device (no device: no split).
static void plan_workspace(aclOpExecutor* e) {
    spfa::AttnParams& p = e->attn;
    auto up256 = [](uint64_t x) { return (x + 255) & ~255ull; };
    const bool per_head = (p.mode == 0 && p.pos_rows == p.Hq && p.G > 1);
    const int g_eff = per_head ? 1 : p.G;
    const long long items = (long long)p.B * (per_head ? p.Hq : p.Hkv) * ((g_eff + p.rows_per_pass - 1) / p.rows_per_pass);
    p.n_split = 1;
    const int n_cu = spfa::device_cu_count();
    long long cap = 0;   // resident workgroups of the kernel this plan selects (no device: 0, no split)
    e->device = current_device_or_none();
    if (e->device >= 0 && p.mode == 1) {   // set-up time: the library-owned scratch of this device gets its address now
        void* unused = nullptr;
        if (spfa::static_scratch(&unused, nullptr) != hipSuccess) (void)hipGetLastError();
    }
    p.grid_cap = 0;
    if (n_cu > 0 && items > 0 && spfa::launch_attention(p, nullptr, &cap) != hipSuccess) cap = 0;
    p.grid_cap = cap;   // phase 2 launches with this: no runtime queries on the launch path
    if (items > 0 && items * 2 <= cap) {
        // One workgroup per CU streams at about half the CU's share of HBM, so fill the idle CUs first; beyond that a
        // second resident workgroup per CU still helps (measured: 256 items 0.150 -> 0.146 ms), more shares per item
        // only add selections and merges (32 items: 8 shares 0.054 ms, 16 shares 0.065 ms).
        long long ns = n_cu / items;
        if (ns < 2) ns = 2;
        if (ns > g_max_split) ns = g_max_split;
        const long long list_pages = p.token_mode ? p.max_entries / p.block_size : p.max_entries;
        if (ns > list_pages) ns = list_pages;   // at least one page per share
        if (ns > 1) p.n_split = (int)ns;
    }
    // Fused op, one pass per (sequence, kv head), a launch of exactly J items per resident workgroup with the workgroup count a
    // multiple of the kv heads: workgroup w then owns items w, w + grid, ... — J sequences of ONE kv head — and scores
    // their J rows in one pass over that head's centroids (the MFMA tile has 16 rows and one pass costs the same for 1
    // or 16 of them), instead of J passes of which J - 1 run beside the page stream and slow it down.  Taken only when
    // the larger key area leaves the residency (and with it the grid) unchanged.  Measured: int8 S=64k 0.780 -> 0.767 ms,
    // configs[2] equal within noise (its second item gains 5 us, the first selection grows by 3 us).  Group passes, which the
    // kernel would take as well (J * R <= 16 rows), lose: GQA 32/8 S=32k 0.241 -> 0.242 ms, S=128k k=64 0.752 -> 0.767 ms —
    // their first selection walks J * R rows with 6 or 8 waves, all of it exposed.
    p.pre_items = 0;
    p.pre_lists = 0;
    const bool one_head = p.rows_per_pass == 1 && p.G == 1;
    // Group passes (one pass per group on the matrix-core streamer): only the SCORES of the workgroup's items are taken
    // in the shared pass; their top-k and page selection stay with the selector waves, beside the page stream.  What
    // leaves the hidden part is the centroid pass itself, whose LDS and L2 traffic slows the streamers of the same CU.
    const bool one_group = p.G > 1 && p.G <= p.rows_per_pass && spfa::attention_uses_mfma(p.rows_per_pass, p.kv_dtype, p.D) &&
                           !p.pse && !p.amask;
    auto try_scored_ahead = [&]() {
        if (!(g_shared_scoring && p.mode == 1 && p.n_split == 1 && cap > 0 && items > cap && items % cap == 0 &&
              (one_head || one_group) && cap % p.Hkv == 0 && (items / cap) * p.rows_per_pass <= (one_head ? 8 : 16)))
            return false;
        spfa::AttnParams trial = p;
        trial.pre_items = (int)(items / cap);
        trial.pre_lists = one_head ? 1 : 0;
        trial.lds_bytes = carve_lds(trial, (size_t)p.rows_per_pass);
        long long cap2 = 0;
        if (trial.lds_bytes <= kMaxLds && spfa::launch_attention(trial, nullptr, &cap2) == hipSuccess && cap2 == cap) {
            trial.grid_cap = cap;
            p = trial;
            return true;
        }
        return false;
    };
    const bool ahead = try_scored_ahead();
    // Batched selection through the queue (one q head per pass, more items than resident workgroups, no exact fit for the
    // static form above): super-items of J consecutive sequences of one kv head, in phases of decreasing J.  A phase
    // holds whole rounds of the grid (one super-item per resident workgroup and round), so every workgroup gets the same
    // number of large super-items first; what does not fill a round goes to the next smaller size, the rest is dealt
    // sequence by sequence and evens the launch out (the last item of each workgroup is shared on top of that).
    // Taken only when the larger selection areas leave the residency unchanged.
    auto try_super_items = [&](spfa::AttnParams& q, long long cap_now) {
        q.sup_items = 0;
        if (!(g_shared_scoring && g_super_items && q.mode == 1 && q.n_split == 1 && cap_now > 0 && items > cap_now && one_head &&
              q.pre_items <= 1 && !q.token_mode && !q.antiq_per_token && q.B > 1))
            return false;
        const char* forced = getenv("SPFA_SUPER_J");   // experiments only: the largest J
        const long long per_round = std::max<long long>(1, cap_now / q.Hkv);   // super-items per kv head in one round
        spfa::AttnParams trial = q;
        int rem = q.B, nph = 0, jmax = 0;
        for (int ph = 0; ph < 4; ++ph) trial.sup_j[ph] = 1, trial.sup_n[ph] = 0;
        for (int J = forced ? atoi(forced) : 8; J >= 1 && rem > 0 && nph < 4; J >>= 1) {
            const bool last = J == 1 || nph == 3;
            const long long n = last ? (rem + J - 1) / J : (rem / (J * per_round)) * per_round;
            if (n <= 0) continue;
            trial.sup_j[nph] = J;
            trial.sup_n[nph] = (int)n;
            rem -= (int)std::min<long long>(rem, n * J);
            jmax = std::max(jmax, J);
            ++nph;
        }
        if (jmax < 2) return false;
        trial.sup_items = jmax;
        trial.pre_items = 0;
        trial.pre_lists = 0;
        trial.lds_bytes = carve_lds(trial, (size_t)q.rows_per_pass);
        long long cap2 = 0;
        trial.grid_cap = 0;
        if (trial.lds_bytes <= kMaxLds && spfa::launch_attention(trial, nullptr, &cap2) == hipSuccess && cap2 == cap_now) {
            trial.grid_cap = cap_now;
            q = trial;
            return true;
        }
        return false;
    };
    const bool supers_wide = !ahead && try_super_items(p, cap);
    // The plain one-q-head kernel exists in two geometries: one workgroup of eight streamer waves per CU (wide: one pass
    // over the centroids per CU at the start, eight items scored ahead) and two workgroups of four (narrow: twice the
    // selector waves).  Wide wins whenever the selector has little to do beside the stream — lists given, every
    // workgroup's items scored ahead, or at most one item per workgroup; a fused launch that pulls more items than that
    // through the queue goes narrow, because one selector wave cannot prepare short lists as fast as eight streamers
    // consume them.
    const bool both = p.rows_per_pass == 1 && !p.token_mode && !p.antiq_per_token &&
                      !spfa::attention_uses_mfma(p.rows_per_pass, p.kv_dtype, p.D);
    // (with batched selection one selector wave feeds eight streamers: wide)
    const bool want_narrow = g_geometry == 2 ||
                             (g_geometry == 0 && p.mode == 1 && cap > 0 && items > cap && !ahead && !supers_wide);
    if (both && want_narrow && n_cu > 0) {
        spfa::AttnParams trial = p;
        trial.narrow = 1;
        trial.pre_items = 0;
        trial.pre_lists = 0;
        trial.sup_items = 0;
        trial.lds_bytes = carve_lds(trial, (size_t)p.rows_per_pass);
        long long cap2 = 0;
        trial.grid_cap = 0;
        if (trial.lds_bytes <= kMaxLds && spfa::launch_attention(trial, nullptr, &cap2) == hipSuccess && cap2 > 0) {
            trial.grid_cap = cap2;
            p = trial;
            cap = cap2;
            p.n_split = 1;
            if (items > 0 && items * 2 <= cap) {   // forced narrow on a small launch: the split as above
                long long ns = n_cu / items;
                if (ns < 2) ns = 2;
                if (ns > g_max_split) ns = g_max_split;
                if (ns > p.max_entries) ns = p.max_entries;
                if (ns > 1) p.n_split = (int)ns;
            }
            if (!try_scored_ahead()) (void)try_super_items(p, cap);
        }
    }
    // Which class (item % 8 = kv head % 8 when Hkv is a multiple of 8) a workgroup's claims start with.  Staying with
    // the own XCD's class keeps that XCD's centroids in its L2, but a kv head whose pages spread over more clusters
    // has longer lists in EVERY sequence, and all of its items would land on one XCD.  Rotate when the centroids of
    // two classes fit the 4 MiB L2 comfortably (or there are none to keep: lists given), measured on GQA 32/8:
    // 0.244 -> 0.239 ms; with 32 kv heads the mean is unchanged, so those keep the affinity.
    const uint64_t cent_bytes_per_class = (uint64_t)((p.Hkv + 7) / 8) * p.C * p.D * 2;
    p.class_step = (p.mode == 0 || 2 * cent_bytes_per_class <= (2u << 20)) ? 1 : 0;
    // Sharing of every workgroup's last item (launches without a split): one board + data area per resident
    // workgroup (at most two per CU): header, page list, q rows of the VALU streamer, 1 + tail_chunks states.
    // The protocol costs about 10 us per workgroup (claims, one more state to merge); it pays when a workgroup has
    // several items to be out of step by, or long ones.
    // Group passes (MFMA streamer) run one workgroup per CU with a few long, unequal items each (the union of a
    // group's lists varies from item to item): measured 2.4 % faster with sharing at 2 items per workgroup.
    const bool mfma = spfa::attention_uses_mfma(p.rows_per_pass, p.kv_dtype, p.D);
    const bool worth = items >= 6ll * n_cu || (items > n_cu && (p.max_entries >= 512 || mfma));
    p.tail_on = (g_tail_sharing && p.n_split == 1 && n_cu > 0 && !p.token_mode && (worth || g_tail_sharing == 2)) ? 1 : 0;
    p.dyn_pages = g_tail_sharing ? 1 : 0;   // spfaSetTailSharing(0) asks for a reproducible summation order
    const uint64_t n_boards = p.tail_on ? (uint64_t)(cap > 0 ? cap : 2ll * n_cu) : 0;   // one per resident workgroup
    const uint64_t state_bytes = (uint64_t)p.rows_per_pass * (p.D + 2) * 4;
    // one state per chunk + the owner's: up to 64 chunks where that stays within 48 KiB per board, else 32 (16 for the
    // wide states of a group pass)
    p.tail_chunks = p.rows_per_pass <= 4 ? (int)std::max<uint64_t>(32, std::min<uint64_t>(64, (48u << 10) / state_bytes)) : 16;
    p.tail_stride = (long long)up256(32 + 144 + (uint64_t)p.max_entries * 8 + (mfma ? 0 : (uint64_t)p.rows_per_pass * p.D * 4) + 16 +
                                     (uint64_t)(p.tail_chunks + 1) * state_bytes);
    // Shared first scoring (fused op with a page-list split): the workgroups of one item score its rows together.
    uint64_t n_groups = 0;
    if (p.mode == 1 && p.n_split > 1 && g_shared_scoring) n_groups = (uint64_t)items;
    p.pg_on = n_groups > 0 ? 1 : 0;
    e->ws_boards_off = kCounterBytes + (p.n_split > 1 ? up256((uint64_t)items * 4) : 0);
    e->ws_pgdone_off = e->ws_boards_off + up256(n_boards * 16);
    e->ws_batch_off = e->ws_pgdone_off + up256(n_groups * 4);   // per-batch arrival counts [B] u32, then maxima [B] u64
    const uint64_t batch_words = (p.mode == 1 && p.max_page_position_length) ? (uint64_t)p.B : 0;
    e->ws_zero_bytes = e->ws_batch_off + up256(batch_words * 4) + up256(batch_words * 8);
    e->ws_parts_off = e->ws_zero_bytes;
    const uint64_t parts = p.n_split > 1 ? up256((uint64_t)items * p.n_split * state_bytes) : 0;
    e->ws_tail_off = e->ws_parts_off + parts;
    e->ws_pgkeys_off = e->ws_tail_off + n_boards * (uint64_t)p.tail_stride;
    e->ws_lengths_off = e->ws_pgkeys_off + n_groups * 16ull * p.Cp * 4;
    e->ws_bytes = e->ws_lengths_off + (e->act_seq.empty() ? 0 : up256((uint64_t)p.B * 8));
}

// Bind the planned workspace layout: the part every launch expects zeroed (counters, arrival words, boards) at
// zero_base, the data areas (partial states, shared page lists, exchanged keys, lengths) at the offsets phase 1 reported
// from data_base.  Both are the caller's workspace unless the zeroed part lives in library-owned memory.  zero: enqueue
// the zeroing of that part first (needed unless it is known to be in the state the previous launch left it in).
static hipError_t bind_workspace(const aclOpExecutor* e, spfa::AttnParams& p, void* zero_base, void* data_base,
                                 hipStream_t stream, bool zero, uint64_t zero_bytes = 0) {
    if ((((uintptr_t)zero_base | (uintptr_t)data_base) & 15) != 0) return hipErrorInvalidValue;   // 64-bit atomics, 16-byte boards
    auto up256 = [](uint64_t x) { return (x + 255) & ~255ull; };
    char* const zb = (char*)zero_base;
    char* const db = (char*)data_base;
    p.counters = (unsigned*)zb;
    p.zero_words = e->ws_zero_bytes / 8;
    p.split_done = (unsigned*)(zb + kCounterBytes);
    p.tail_boards = zb + e->ws_boards_off;
    p.pg_done = (unsigned*)(zb + e->ws_pgdone_off);
    p.batch_arrive = (unsigned*)(zb + e->ws_batch_off);
    p.batch_max = (unsigned long long*)(zb + e->ws_batch_off + up256((uint64_t)p.B * 4));
    p.split_partials = (float*)(db + e->ws_parts_off);
    p.tail_data = db + e->ws_tail_off;
    p.pg_keys = (uint32_t*)(db + e->ws_pgkeys_off);
    // (a slot of the library is zeroed as a whole the first time it is used: later launches with other plans rely on it)
    return zero ? spfa::launch_zero(zero_base, zero_bytes ? zero_bytes : e->ws_zero_bytes, nullptr, 0, stream) : hipSuccess;
}

static bool stream_is_capturing(hipStream_t stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

// Library-owned counters.  The zeroed part of a workspace is small (a few KiB) but a caller's buffer cannot be trusted to
// hold zeros, so a launch on it is two kernels.  Launches on ONE stream are ordered, and every launch leaves its
// counters zeroed: so each (device, stream) pair is given a 64 KiB slot of the library's own device array, zeroed once
// and then kept clean by the launches themselves, and a launch whose zeroed part fits runs as one kernel with only its
// data areas in the caller's workspace.  Not under stream capture (a graph may be replayed on any stream), not on the
// per-thread default stream handle (one handle, many streams); when the slots are used up the caller's workspace and
// the zeroing kernel are used as before.
struct StreamSlot { uintptr_t key; bool zeroed; };
static StreamSlot g_slots[spfa::kMaxScratchDevices][spfa::kScratchSlots];
static bool g_noalloc_zeroed[spfa::kMaxScratchDevices];
static std::mutex g_slot_mutex;

// The slot of (current device, stream) as a device address, or nullptr.  *needs_zero: first use of the slot.
static void* stream_slot(hipStream_t stream, uint64_t zero_bytes, bool* needs_zero) {
    if (!g_stream_slots || zero_bytes > spfa::kScratchSlotBytes) return nullptr;
    if (stream == hipStreamPerThread || stream == hipStreamLegacy) return nullptr;
    void* base = nullptr;
    int dev = -1;
    if (spfa::static_scratch(&base, &dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (dev < 0 || dev >= spfa::kMaxScratchDevices) return nullptr;
    const uintptr_t key = (uintptr_t)stream + 1;   // 0 marks a free slot; the null stream is a stream like any other
    std::lock_guard<std::mutex> lock(g_slot_mutex);
    for (int i = 0; i < spfa::kScratchSlots; ++i) {
        StreamSlot& s = g_slots[dev][i];
        if (s.key == 0) { s.key = key; s.zeroed = false; }
        if (s.key == key) {
            *needs_zero = !s.zeroed;
            s.zeroed = true;   // the zeroing is enqueued on this stream in front of the first launch that uses the slot
            return (char*)base + spfa::kScratchSlotsOff + (size_t)i * spfa::kScratchSlotBytes;
        }
    }
    // Every slot belongs to some stream: take over the slot of a stream that is idle right now — everything it was
    // given has run to completion, and every launch leaves its slot clean.  (A stream that no longer exists answers the
    // query with an error and keeps its slot: nothing is known about work it may still have had in flight.)
    for (int i = 0; i < spfa::kScratchSlots; ++i) {
        StreamSlot& s = g_slots[dev][i];
        if (!s.zeroed) continue;
        if (hipStreamQuery((hipStream_t)(s.key - 1)) == hipSuccess) {
            s.key = key;
            *needs_zero = false;
            return (char*)base + spfa::kScratchSlotsOff + (size_t)i * spfa::kScratchSlotBytes;
        }
        (void)hipGetLastError();
    }
    return nullptr;
}
// a failed enqueue leaves the slot in an unknown state: zero it again next time
static void stream_slot_failed(hipStream_t stream) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= spfa::kMaxScratchDevices) return;
    const uintptr_t key = (uintptr_t)stream + 1;
    std::lock_guard<std::mutex> lock(g_slot_mutex);
    for (int i = 0; i < spfa::kScratchSlots; ++i)
        if (g_slots[dev][i].key == key) g_slots[dev][i].zeroed = false;
}

// Where the zeroed part of this launch lives and whether it has to be zeroed first.
struct ZeroPlace { void* base; bool zero; bool slot; };
static ZeroPlace place_zeroed_part(aclOpExecutor* e, void* workspace, hipStream_t stream) {
    // Under capture nothing runs now: the bookkeeping below describes enqueued work, so it is neither consulted beyond
    // "this executor owns the workspace and has run on it" nor updated, and a replay zeroes for itself otherwise.
    if (stream_is_capturing(stream)) {
        const bool known_clean = e->repeatable && e->persistent_ws && e->clean_ws == workspace;
        return ZeroPlace{workspace, !known_clean, false};
    }
    if (e->repeatable && e->persistent_ws) {
        const bool known_clean = e->clean_ws == workspace;
        e->clean_ws = workspace;
        return ZeroPlace{workspace, !known_clean, false};
    }
    bool needs_zero = false;
    if (void* slot = stream_slot(stream, e->ws_zero_bytes, &needs_zero)) return ZeroPlace{slot, needs_zero, true};
    return ZeroPlace{workspace, true, false};
}
static void launch_outcome(aclOpExecutor* e, const ZeroPlace& z, hipStream_t stream, hipError_t err) {
    if (err == hipSuccess) return;
    if (z.slot) stream_slot_failed(stream);
    else if (e->repeatable) e->clean_ws = nullptr;
}

// Phase 2 runs on the device phase 1 planned for (grid size, LDS limit and residency belong to that device).
static aclnnStatus check_device(const aclOpExecutor* e) {
    const int now = current_device_or_none();
    REQUIRE(e->device < 0 || now == e->device, "executor was planned on device %d but device %d is current", e->device, now);
    return ACLNN_SUCCESS;
}

#define REJECT_PRESENT(t, name) \
    REQUIRE(absent(t), "%s is not supported on the sparse paged path (must be empty)", name)

aclnnStatus aclnnSparsePagedAttentionGetWorkspaceSize(
    const aclTensor* query, const aclTensorList* key, const aclTensorList* value, const aclTensor* pseShift,
    const aclTensor* attenMask, const aclIntArray* actualSeqLengths, const aclTensor* dequantScale1,
    const aclTensor* quantScale1, const aclTensor* dequantScale2, const aclTensor* quantScale2,
    const aclTensor* quantOffset2, const aclTensor* antiquantScale, const aclTensor* antiquantOffset,
    const aclTensor* blocktable, const aclTensor* kvPaddingSize, const aclTensor* blockposition, int64_t numHeads,
    double scaleValue, char* inputLayout, int64_t numKeyValueHeads, int64_t blockSize, int64_t /*innerPrecise*/,
    const aclTensor* attentionOut, uint64_t* workspaceSize, aclOpExecutor** executor) {
    // one-shot: taken before anything can fail, so a rejected call never leaves a device pointer armed for the next one
    const long long* const act_seq_dev = g_act_seq_dev;
    const int64_t act_seq_dev_n = g_act_seq_dev_n;
    g_act_seq_dev = nullptr;
    g_act_seq_dev_n = 0;
    const int token_positions = g_token_positions;
    g_token_positions = 0;
    const AttentionExtras extras = take_attention_extras();
    REQUIRE_PTR(workspaceSize, "workspaceSize");
    REQUIRE_PTR(executor, "executor");
    REJECT_PRESENT(dequantScale1, "dequantScale1");   // int8 queries are not part of this path
    REJECT_PRESENT(quantScale1, "quantScale1");
    REJECT_PRESENT(dequantScale2, "dequantScale2");
    (void)kvPaddingSize;   // left padding only exists for contiguous KV: with a block table the reference drops it too
                           // (…_tiling_check.cc:590-594)
    aclOpExecutor* e = new aclOpExecutor();
    e->op = kOpAttention;
    spfa::AttnParams& p = e->attn;
    p.mode = 0;
    aclnnStatus st = plan_attention(p, query, key, value, antiquantScale, antiquantOffset, blocktable, numHeads,
                                    scaleValue, inputLayout, numKeyValueHeads, blockSize, attentionOut, -1, extras.antiq_mode);
    auto bail = [&](aclnnStatus s) { delete e; return s; };
    if (st != ACLNN_SUCCESS) return bail(st);
    st = check_tensor(blockposition, "blockposition", 3, ACL_INT32);
    if (st != ACLNN_SUCCESS) return bail(st);
    const int rows = (int)blockposition->dims[1];
    if (blockposition->dims[0] < p.B || (rows != p.Hq && rows != p.Hkv) || blockposition->dims[2] < 1)
        return bail(fail(ACLNN_ERR_PARAM_INVALID, "blockposition must be [B, numHeads | numKeyValueHeads, P]"));
    p.block_position = (const int32_t*)data_ptr(blockposition);
    p.pos_rows = rows;
    p.P = (int)blockposition->dims[2];
    p.max_entries = p.P;
    if (token_positions) {
        // token-level lists (the variant sketched in the reference's tests/ifa_v5_case_tkp.py:53-54): one 8-byte entry per
        // token in LDS, padded to whole virtual pages of blockSize entries
        const int lpt_tok = p.D * (p.kv_dtype == spfa::kI8 ? 1 : 2) / 16;
        if (p.D != 64 && p.D != 128)
            return bail(fail(ACLNN_ERR_PARAM_INVALID, "token-level positions support head dims 64 and 128, got %d", p.D));
        (void)lpt_tok;
        if (!absent(pseShift) || !absent(attenMask))
            return bail(fail(ACLNN_ERR_PARAM_INVALID, "pseShift / attenMask are not supported with token-level positions"));
        p.token_mode = 1;
        // the LDS area holds a window of the list (whole virtual pages); longer lists are streamed window by window
        // (plan_lds settles the window)
        p.max_entries = round_up(p.P, p.block_size);
    }
    p.act_seq_uniform = -1;
    if (act_seq_dev) {   // extension: lengths already on the device, no host round trip
        if (act_seq_dev_n < p.B)
            return bail(fail(ACLNN_ERR_PARAM_INVALID, "device actualSeqLengths has %lld values < batch %d",
                             (long long)act_seq_dev_n, p.B));
        p.act_seq = act_seq_dev;
    } else if (actualSeqLengths && !actualSeqLengths->v.empty()) {
        const auto& v = actualSeqLengths->v;
        if (v.size() != 1 && (int)v.size() < p.B)
            return bail(fail(ACLNN_ERR_PARAM_INVALID, "actualSeqLengths has %zu values, expected 1 or batch %d", v.size(), p.B));
        bool uniform = true;
        for (int b = 1; b < p.B && v.size() > 1; ++b) uniform = uniform && v[b] == v[0];
        if (uniform) {
            p.act_seq_uniform = v[0];
        } else {
            e->act_seq.assign(v.begin(), v.begin() + p.B);
        }
    }
    st = plan_extras(p, query, pseShift, attenMask, quantScale2, quantOffset2, attentionOut);
    if (st != ACLNN_SUCCESS) return bail(st);
    st = plan_oneshot_extras(p, extras);
    if (st != ACLNN_SUCCESS) return bail(st);
    st = plan_lds(p);
    if (st != ACLNN_SUCCESS) return bail(st);
    plan_workspace(e);
    *workspaceSize = e->ws_bytes;
    *executor = e;
    return ACLNN_SUCCESS;
}

aclnnStatus aclnnSparsePagedAttention(void* workspace, uint64_t workspaceSize, aclOpExecutor* executor,
                                      const aclrtStream stream) {
    REQUIRE_PTR(executor, "executor");
    REQUIRE(executor->op == kOpAttention, "executor was not created by aclnnSparsePagedAttentionGetWorkspaceSize");
    spfa::AttnParams p = executor->attn;
    if (workspace == nullptr || workspaceSize < executor->ws_bytes) {
        const unsigned long long need = executor->ws_bytes;
        if (!executor->repeatable) delete executor;
        return fail(ACLNN_ERR_PARAM_INVALID, "workspace of %llu bytes required", need);
    }
    {
        const aclnnStatus ds = check_device(executor);
        if (ds != ACLNN_SUCCESS) { if (!executor->repeatable) delete executor; return ds; }
    }
    const ZeroPlace zp = place_zeroed_part(executor, workspace, (hipStream_t)stream);
    hipError_t err = bind_workspace(executor, p, zp.base, workspace, (hipStream_t)stream, zp.zero,
                                    zp.slot ? spfa::kScratchSlotBytes : 0);
    if (err == hipSuccess && !executor->act_seq.empty()) {
        // Per-batch lengths travel to the workspace as kernel arguments: copied when the launch is enqueued (or
        // captured into a graph), so no host memory has to outlive this call.
        long long* lengths = (long long*)((char*)workspace + executor->ws_lengths_off);
        err = spfa::launch_store_lengths(lengths, executor->act_seq.data(), p.B, (hipStream_t)stream);
        p.act_seq = lengths;
    }
    if (err == hipSuccess) err = spfa::launch_attention(p, (hipStream_t)stream);
    launch_outcome(executor, zp, (hipStream_t)stream, err);
    return finish(executor, err, "sparse_paged_attention launch");
}

aclnnStatus aclnnSparsePagedFusionAttentionGetWorkspaceSize(
    const aclTensor* query, const aclTensorList* key, const aclTensorList* value, const aclTensor* pseShift,
    const aclTensor* attenMask, const aclIntArray* /*actualSeqLengths: tiling hint only*/, const aclTensor* dequantScale1,
    const aclTensor* quantScale1, const aclTensor* dequantScale2, const aclTensor* quantScale2,
    const aclTensor* quantOffset2, const aclTensor* antiquantScale, const aclTensor* antiquantOffset,
    const aclTensor* blocktable, const aclTensor* kvPaddingSize, const aclTensor* l1_cent, const aclTensor* block_ids,
    const aclTensor* total_seq_len, int64_t numHeads, double scaleValue, char* inputLayout, int64_t numKeyValueHeads,
    int64_t blockSize, int64_t /*innerPrecise*/, const aclTensor* block_position, const aclTensor* page_position_length,
    const aclTensor* max_page_position_length, const aclTensor* attentionOut, uint64_t* workspaceSize,
    aclOpExecutor** executor) {
    const SelectionKnobs knobs = take_selection_knobs();
    const AttentionExtras extras = take_attention_extras();
    REQUIRE_PTR(workspaceSize, "workspaceSize");
    REQUIRE_PTR(executor, "executor");
    REJECT_PRESENT(dequantScale1, "dequantScale1");   // int8 queries are not part of this path
    REJECT_PRESENT(quantScale1, "quantScale1");
    REJECT_PRESENT(dequantScale2, "dequantScale2");
    (void)kvPaddingSize;   // left padding only exists for contiguous KV: with a block table the reference drops it too
                           // (…_tiling_check.cc:590-594)
    int Hq = 0, Hkv = 0, D = 0, C = 0;
    CHECK(check_qcent(query, l1_cent, &Hq, &Hkv, &D, &C));
    int B = 0, kv_page_len = 0, max_page = 0;
    CHECK(check_page_tables(block_ids, blocktable, total_seq_len, Hkv, &B, &kv_page_len, &max_page));
    aclOpExecutor* e = new aclOpExecutor();
    e->op = kOpFusion;
    spfa::AttnParams& p = e->attn;
    p.mode = 1;
    auto bail = [&](aclnnStatus s) { delete e; return s; };
    aclnnStatus st = plan_attention(p, query, key, value, antiquantScale, antiquantOffset, blocktable, numHeads,
                                    scaleValue, inputLayout, numKeyValueHeads, blockSize, attentionOut, B, extras.antiq_mode);
    if (st != ACLNN_SUCCESS) return bail(st);
    if (p.Hkv != Hkv || p.D != D)
        return bail(fail(ACLNN_ERR_PARAM_INVALID, "l1_cent [%d, %d, %d] does not match numKeyValueHeads %d / head dim %d",
                         Hkv, C, D, p.Hkv, p.D));
    const int k = (int)knobs.topk;
    if (k > C) return bail(fail(ACLNN_ERR_PARAM_INVALID, "top-k %d exceeds the cluster count %d", k, C));
    p.cent = data_ptr(l1_cent);
    p.block_ids = (const int32_t*)data_ptr(block_ids);
    p.seq_len = (const int32_t*)data_ptr(total_seq_len);
    p.C = C; p.Cp = round_up(C, 128); p.k = k; p.kv_page_len = kv_page_len;
    p.sink_recent = knobs.sink_recent; p.cap_pages = knobs.page_cap;
    p.P = max_page;
    if (!absent(block_position)) {
        st = check_tensor(block_position, "block_position", 3, ACL_INT32);
        if (st != ACLNN_SUCCESS) return bail(st);
        if (block_position->dims[0] < B || block_position->dims[1] != p.Hq || block_position->dims[2] < 1)
            return bail(fail(ACLNN_ERR_PARAM_INVALID, "block_position must be [B, numHeads, P]"));
        p.page_position = (int32_t*)data_ptr(block_position);
        p.P = (int)block_position->dims[2];
    }
    if (!absent(page_position_length)) {
        st = check_tensor(page_position_length, "page_position_length", 3, ACL_INT32);
        if (st != ACLNN_SUCCESS) return bail(st);
        if (page_position_length->dims[0] < B || page_position_length->dims[1] != p.Hq ||
            page_position_length->dims[2] != spfa::kLenPad)
            return bail(fail(ACLNN_ERR_PARAM_INVALID, "page_position_length must be [B, numHeads, 8]"));
        p.page_position_length = (int32_t*)data_ptr(page_position_length);
    }
    if (!absent(max_page_position_length)) {
        st = check_tensor(max_page_position_length, "max_page_position_length", 2, ACL_INT64);
        if (st != ACLNN_SUCCESS) return bail(st);
        if (max_page_position_length->dims[0] < B || max_page_position_length->dims[1] != spfa::kLenPad)
            return bail(fail(ACLNN_ERR_PARAM_INVALID, "max_page_position_length must be [B, 8]"));
        p.max_page_position_length = (unsigned long long*)data_ptr(max_page_position_length);
    }
    p.max_entries = max_page;   // the union of a group's lists never exceeds the logical page count
    st = plan_extras(p, query, pseShift, attenMask, quantScale2, quantOffset2, attentionOut);
    if (st != ACLNN_SUCCESS) return bail(st);
    st = plan_oneshot_extras(p, extras);
    if (st != ACLNN_SUCCESS) return bail(st);
    st = plan_lds(p);
    if (st != ACLNN_SUCCESS) return bail(st);
    plan_workspace(e);
    *workspaceSize = e->ws_bytes;
    *executor = e;
    return ACLNN_SUCCESS;
}

aclnnStatus aclnnSparsePagedFusionAttention(void* workspace, uint64_t workspaceSize, aclOpExecutor* executor,
                                            const aclrtStream stream) {
    REQUIRE_PTR(executor, "executor");
    REQUIRE(executor->op == kOpFusion, "executor was not created by aclnnSparsePagedFusionAttentionGetWorkspaceSize");
    spfa::AttnParams p = executor->attn;
    if (workspace == nullptr || workspaceSize < executor->ws_bytes) {
        const unsigned long long need = executor->ws_bytes;
        if (!executor->repeatable) delete executor;
        return fail(ACLNN_ERR_PARAM_INVALID, "workspace of %llu bytes required", need);
    }
    {
        const aclnnStatus ds = check_device(executor);
        if (ds != ACLNN_SUCCESS) { if (!executor->repeatable) delete executor; return ds; }
    }
    const ZeroPlace zp = place_zeroed_part(executor, workspace, (hipStream_t)stream);
    hipError_t err = bind_workspace(executor, p, zp.base, workspace, (hipStream_t)stream, zp.zero,
                                    zp.slot ? spfa::kScratchSlotBytes : 0);
    if (err == hipSuccess) err = spfa::launch_attention(p, (hipStream_t)stream);
    launch_outcome(executor, zp, (hipStream_t)stream, err);
    return finish(executor, err, "sparse_paged_fusion_attention launch");
}

// ------------------------------------------------------------------------------------------------
// AssignBlockIds (no counterpart in the reference: the producer of block_ids)
// ------------------------------------------------------------------------------------------------
aclnnStatus aclnnAssignBlockIdsGetWorkspaceSize(const aclTensorList* key, const aclTensor* blockTable,
                                                const aclTensor* seqLen, const aclTensor* firstNewPage,
                                                const aclTensor* l1Cent, const aclTensor* antiquantScale,
                                                const aclTensor* antiquantOffset, int64_t blockSize, int64_t maxNewPages,
                                                const aclTensor* blockIdsInout, uint64_t* workspaceSize,
                                                aclOpExecutor** executor) {
    REQUIRE_PTR(workspaceSize, "workspaceSize");
    REQUIRE_PTR(executor, "executor");
    REQUIRE_PTR(l1Cent, "l1_cent");
    REQUIRE(l1Cent->dtype == ACL_FLOAT16 || l1Cent->dtype == ACL_BF16, "l1_cent must be float16 or bfloat16");
    REQUIRE(l1Cent->dims.size() == 3 && is_contiguous(l1Cent) && l1Cent->data, "l1_cent must be a contiguous [Hkv, C, D] tensor");
    const int Hkv = (int)l1Cent->dims[0], C = (int)l1Cent->dims[1], D = (int)l1Cent->dims[2];
    REQUIRE(Hkv > 0 && C > 0 && C <= 16384, "l1_cent: %d heads, %d clusters (1 .. 16384)", Hkv, C);
    REQUIRE(D == 32 || D == 64 || D == 128 || D == 256 || D == 512, "head dim %d: must be 32, 64, 128, 256 or 512", D);
    int B = 0, kv_page_len = 0, max_page = 0;
    CHECK(check_page_tables(blockIdsInout, blockTable, seqLen, Hkv, &B, &kv_page_len, &max_page));
    CHECK(check_tensor(firstNewPage, "first_new_page", 1, ACL_INT32));
    REQUIRE(firstNewPage->dims[0] >= B, "first_new_page has %lld values < batch %d", (long long)firstNewPage->dims[0], B);
    REQUIRE(key && key->items.size() == 1, "key must be a one-tensor list holding the paged K cache");
    const aclTensor* kc = key->items[0];
    REQUIRE(is_contiguous(kc) && kc->data, "key cache must be contiguous device memory");
    const bool quant = kc->dtype == ACL_INT8;
    REQUIRE(quant || kc->dtype == l1Cent->dtype, "key cache dtype must equal l1_cent's dtype or be int8");
    const int es = quant ? 1 : 2;
    REQUIRE(blockSize > 0 && blockSize <= 512, "blockSize %lld out of (0, 512]", (long long)blockSize);
    const int bs = (int)blockSize;
    const auto& kd = kc->dims;
    bool bsh;
    if (kd.size() == 3 && kd[1] == bs && kd[2] == (int64_t)Hkv * D) bsh = true;
    else if (kd.size() == 4 && kd[1] == bs && kd[2] == Hkv && kd[3] == D) bsh = true;
    else if (kd.size() == 4 && kd[1] == Hkv && kd[2] == bs && kd[3] == D) bsh = false;
    else return fail(ACLNN_ERR_PARAM_INVALID, "key cache must be [blocks,bs,Hkv*D], [blocks,bs,Hkv,D] or [blocks,Hkv,bs,D]");
    REQUIRE(kd[0] > 0 && kd[0] <= 0x7fffffff, "key cache has no blocks");
    const int lpt = D * es / 16;
    REQUIRE(lpt >= 4 && lpt <= 64, "head dim %d with %d-byte elements is not supported (row bytes 64 .. 1024)", D, es);
    aclOpExecutor* e = new aclOpExecutor();
    e->op = kOpAssign;
    spfa::AssignParams& p = e->assign;
    auto bail = [&](aclnnStatus s) { delete e; return s; };
    if (quant) {
        if (absent(antiquantScale)) return bail(fail(ACLNN_ERR_PARAM_INVALID, "int8 key cache needs antiquantScale"));
        const bool per_tensor = antiquantScale->dims.size() == 1 && antiquantScale->dims[0] == 2;
        const long long want = per_tensor ? 2 : 2ll * Hkv * D;
        if (antiquantScale->dtype != l1Cent->dtype || numel(antiquantScale) != want || !is_contiguous(antiquantScale))
            return bail(fail(ACLNN_ERR_PARAM_INVALID, "antiquantScale must be [2, Hkv, D] or [2] in l1_cent's dtype"));
        if (!absent(antiquantOffset) && (antiquantOffset->dtype != l1Cent->dtype || numel(antiquantOffset) != want ||
                                         !is_contiguous(antiquantOffset)))
            return bail(fail(ACLNN_ERR_PARAM_INVALID, "antiquantOffset must have antiquantScale's shape and dtype"));
        p.antiq_per_tensor = per_tensor ? 1 : 0;
        p.antiq_scale = data_ptr(antiquantScale);
        p.antiq_offset = absent(antiquantOffset) ? nullptr : data_ptr(antiquantOffset);
    } else if (!absent(antiquantScale) || !absent(antiquantOffset)) {
        return bail(fail(ACLNN_ERR_PARAM_INVALID, "antiquant parameters given but the key cache is not int8"));
    }
    p.kcache = data_ptr(kc);
    if (bsh) {
        p.blk_stride = (long long)bs * Hkv * D * es; p.tok_stride = (long long)Hkv * D * es; p.head_stride = (long long)D * es;
    } else {
        p.blk_stride = (long long)Hkv * bs * D * es; p.head_stride = (long long)bs * D * es; p.tok_stride = (long long)D * es;
    }
    p.kv_dtype = to_dtype(kc->dtype);
    p.n_blocks = (int)kd[0];
    p.cent = data_ptr(l1Cent);
    p.qk_dtype = to_dtype(l1Cent->dtype);
    p.block_table = (const int32_t*)data_ptr(blockTable);
    p.seq_len = (const int32_t*)data_ptr(seqLen);
    p.first_new = (const int32_t*)data_ptr(firstNewPage);
    p.B = B; p.Hkv = Hkv; p.D = D; p.C = C; p.Cp = round_up(C, 128); p.block_size = bs; p.max_page = max_page;
    p.max_new = (maxNewPages <= 0 || maxNewPages > max_page) ? max_page : (int)maxNewPages;
    p.kv_page_len = kv_page_len;
    p.block_ids = (int32_t*)data_ptr(blockIdsInout);
    int rows = 16;   // as many rows per centroid pass as the keys leave room for
    while (rows > 1 && spfa::assign_lds_bytes(rows, p.Cp, D) > 150 * 1024) --rows;
    p.rows_per_pass = rows;
    p.lds_bytes = spfa::assign_lds_bytes(rows, p.Cp, D);
    if (p.lds_bytes > kMaxLds) return bail(fail(ACLNN_ERR_PARAM_INVALID, "%d clusters need %zu bytes of LDS (> %zu)", C, p.lds_bytes, kMaxLds));
    e->device = current_device_or_none();
    *workspaceSize = 0;
    *executor = e;
    return ACLNN_SUCCESS;
}

aclnnStatus aclnnAssignBlockIds(void*, uint64_t, aclOpExecutor* executor, aclrtStream stream) {
    REQUIRE_PTR(executor, "executor");
    REQUIRE(executor->op == kOpAssign, "executor was not created by aclnnAssignBlockIdsGetWorkspaceSize");
    {
        const aclnnStatus ds = check_device(executor);
        if (ds != ACLNN_SUCCESS) { if (!executor->repeatable) delete executor; return ds; }
    }
    return finish(executor, spfa::launch_assign(executor->assign, (hipStream_t)stream), "assign_block_ids launch");
}

// Phase 2 of SparsePagedFusionAttention for callers that cannot allocate device memory: the workspace is a fixed array
// inside the library's code object (one per GPU, 32 MiB) instead of a caller buffer.  Nothing is allocated or freed, on
// the device or through any allocating runtime call.  The array has a fixed part for what launches expect zeroed (its
// first MiB) and a fixed part for data (from 2 MiB on): every launch leaves its own zeroed part zeroed and writes data
// only behind 2 MiB, so between launches the first MiB holds zeros whatever ran before — launches of other executors with
// larger or smaller plans, replays of captured launches — and no host-side record of earlier launches is consulted.  The
// first MiB is zeroed once per device by the first call that is not being captured (a captured call made before that
// records the zeroing of its own part into the graph).  Calls that overlap on one GPU share the array and race.
aclnnStatus aclnnSparsePagedFusionAttentionNoAlloc(aclOpExecutor* executor, const aclrtStream stream) {
    REQUIRE_PTR(executor, "executor");
    REQUIRE(executor->op == kOpFusion, "executor was not created by aclnnSparsePagedFusionAttentionGetWorkspaceSize");
    auto drop = [&]() { if (!executor->repeatable) delete executor; };
    const uint64_t data_bytes = executor->ws_bytes - executor->ws_zero_bytes;
    if (executor->ws_zero_bytes > spfa::kScratchZeroBytes || data_bytes > spfa::kStaticScratchBytes - spfa::kScratchDataOff) {
        const unsigned long long need = executor->ws_bytes;
        drop();
        return fail(ACLNN_ERR_PARAM_INVALID,
                    "this launch needs %llu bytes of workspace, the built-in scratch holds %llu (%llu of them for counters): "
                    "plan it with spfaSetTailSharing(0) (and spfaSetMaxSplit(1) for small batches), or pass a workspace to "
                    "aclnnSparsePagedFusionAttention", need,
                    (unsigned long long)(spfa::kStaticScratchBytes - spfa::kScratchDataOff + spfa::kScratchZeroBytes),
                    (unsigned long long)spfa::kScratchZeroBytes);
    }
    {
        const aclnnStatus ds = check_device(executor);
        if (ds != ACLNN_SUCCESS) { drop(); return ds; }
    }
    void* scratch_ptr = nullptr;
    int dev = -1;
    hipError_t err = spfa::static_scratch(&scratch_ptr, &dev);
    if (err != hipSuccess) { drop(); return hip_fail(err, "address of the built-in scratch"); }
    if (dev < 0 || dev >= spfa::kMaxScratchDevices) { drop(); return fail(ACLNN_ERR_RUNTIME_ERROR, "device %d out of range", dev); }
    spfa::AttnParams p = executor->attn;
    const bool capturing = stream_is_capturing((hipStream_t)stream);
    const bool first = !__atomic_load_n(&g_noalloc_zeroed[dev], __ATOMIC_ACQUIRE);
    if (first && !capturing)   // once per device: the whole fixed zeroed part
        err = spfa::launch_zero(scratch_ptr, spfa::kScratchZeroBytes, nullptr, 0, (hipStream_t)stream);
    // the data offsets of the plan count from the start of a workspace whose first ws_zero_bytes hold the zeroed part
    char* const data_base = (char*)scratch_ptr + spfa::kScratchDataOff - executor->ws_zero_bytes;
    if (err == hipSuccess) err = bind_workspace(executor, p, scratch_ptr, data_base, (hipStream_t)stream, first && capturing);
    if (err == hipSuccess) err = spfa::launch_attention(p, (hipStream_t)stream);
    if (!capturing) __atomic_store_n(&g_noalloc_zeroed[dev], err == hipSuccess, __ATOMIC_RELEASE);
    return finish(executor, err, "sparse_paged_fusion_attention launch (built-in scratch)");
}

}  // extern "C"
