// Plain-old-data launch descriptors shared by the host planner (aclnn_ops.cpp) and the
// gfx950 kernels (spfa_kernels.hip).  They play the role of the reference's tiling structs
// (CentSelectTilingData, cent_select_tiling.h:5-27; SparsePagedFusionAttentionTilingData,
// sparse_paged_fusion_attention_tiling.h:131-142): the only channel host -> kernel.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace spfa {

constexpr int kSentinel = 0x7fffffff;   // cent_select.cpp:406
constexpr int kLenPad = 8;              // page_position_length / max_page_position_length lanes
constexpr int kThreads = 256;           // 4 wave64 per workgroup
constexpr int kCentWaves = 8;           // waves per workgroup of the standalone scoring kernels: one q row each after
                                        // the shared centroid pass (8 rows per pass at C = 2048)

enum DType : int { kF16 = 0, kBF16 = 1, kI8 = 2, kF32 = 3 };

// ComputeCent / CentSelect (no attention).  One workgroup per (group of sequences, kv head).
struct CentParams {
    const void* q;             // [Bq, Hq, D]   f16 | bf16
    const void* cent;          // [Hkv, C, D]   same dtype
    int qk_dtype;              // kF16 | kBF16
    int B, Hq, Hkv, G, D, C, k;
    int rows_per_pass;         // q rows scored per centroid pass (LDS budget of the keys, <= 16)
    int score_waves;           // waves of the workgroup that share the centroid pass (one LDS tile each)
    int seq_per_group;         // sequences whose rows share one pass (rows_per_pass / G when G fits, else 1)
    int Cp;                    // C rounded up to 64
    // mode 0: compute_cent (sorted indices out)
    int32_t* indices;          // [B, Hq, k]
    // mode 1: cent_select
    const int32_t* block_ids;  // [Hkv, kv_page_len]
    const int32_t* block_table;// [max_batch, max_page]
    const int32_t* seq_len;    // [B]
    int kv_page_len, max_page, block_size;
    int P;                     // slots per page_position row
    int sink_recent, cap_pages;
    int32_t* page_position;        // [B, Hq, P]
    int32_t* page_position_length; // [B, Hq, 8]
    unsigned long long* max_page_position_length; // [B, 8] (int64, non negative)
    // per-sequence arrival counts and running maxima in zeroed memory that the launch leaves zeroed (the row that
    // completes a sequence's Hq rows stores the maximum and clears both words); null: atomicMax straight into
    // max_page_position_length, which then has to be zeroed before the launch
    unsigned* batch_arrive;        // [B]
    unsigned long long* batch_max; // [B]
    size_t lds_bytes;
};

// SelectPosition.  One wave per (batch, q head).
struct SelectParams {
    const int32_t* block_ids;   // [Hkv, kv_page_len]
    const int32_t* block_table; // [max_batch, max_page]
    const int32_t* seq_len;     // [B]
    const int32_t* indices;     // [B, Hq, k]
    int B, Hq, Hkv, G, k, kv_page_len, max_page, block_size, P;
    int32_t* page_position;        // [B, Hq, P]
    int32_t* page_position_length; // [B, Hq, 8]
    int32_t* block_table_gather;   // [B, Hq, max_page]
};

// AssignBlockIds: cluster ids of newly filled KV pages.  One workgroup per (kv head, group of rows_per_pass new pages).
struct AssignParams {
    const void* kcache;        // BSH [blocks, bs, Hkv, D] | BNSD [blocks, Hkv, bs, D]
    long long blk_stride, tok_stride, head_stride;   // bytes
    int kv_dtype, n_blocks;
    const void* cent;          // [Hkv, C, D] f16 | bf16
    int qk_dtype;              // dtype of the centroids (the page means are rounded to it)
    const void* antiq_scale;   // int8 K: [2, Hkv*D] or [2] in the centroid dtype (key parameters first), else null
    const void* antiq_offset;  // same shape, or null
    int antiq_per_tensor;
    const int32_t* block_table;// [>=B, max_page]
    const int32_t* seq_len;    // [B] tokens in the sequence
    const int32_t* first_new;  // [B] first logical page without a cluster id yet
    int B, Hkv, D, C, Cp, block_size, max_page;
    int max_new;               // most new pages per sequence one call looks at
    int rows_per_pass;         // new pages scored per centroid pass (<= 16; LDS budget of the keys)
    int kv_page_len;
    int32_t* block_ids;        // [Hkv, kv_page_len] in/out
    size_t lds_bytes;
};

// SparsePagedAttention (mode 0) and SparsePagedFusionAttention (mode 1).
struct AttnParams {
    int mode;                  // 0: page lists given (block_position), 1: fused cent-select first
    const void* q;             // [B, Hq, D]
    void* out;                 // [B, Hq, D]
    int q_dtype, out_dtype, kv_dtype;
    const void* kcache;        // BSH [blocks, bs, Hkv, D] | BNSD [blocks, Hkv, bs, D]
    const void* vcache;
    long long blk_stride, tok_stride, head_stride;   // bytes
    int n_blocks;
    const int32_t* block_table;// [>=B, max_page]
    int max_page;
    int B, Hq, Hkv, G, D, block_size;
    float scale;
    const void* antiq_scale;   // [2, Hkv*D] in q dtype, or null
    const void* antiq_offset;  // [2, Hkv*D] in q dtype, or null
    int antiq_per_tensor;      // layout of antiq_scale / antiq_offset: 0 per channel [2, Hkv, D], 1 per tensor [2] (one
                               // key and one value parameter), 2 per kv head [2, Hkv]
    // 1: per-token parameters (…Us2.h:2039-2041, 3711-3822): antiq_scale / antiq_offset are float32 [2, B, antiq_s], one
    // key and one value parameter per column of the gathered sequence (the columns of pse / amask); every q head runs
    // as a pass of its own
    int antiq_per_token;
    long long antiq_s;
    int antiq_tok_heads;       // per-token parameters with a row per kv head, float32 [2, B, Hkv, antiq_s]: Hkv, else 0
    // Additive score bias and boolean mask over the columns of the gathered sequence (column = list slot * block_size +
    // token in page), …Us2.h:3366-3404.  Either one makes every q head run as a pass of its own.
    const void* pse;           // [pse_b, Hq, 1, pse_s] in q dtype, or null
    int pse_b;                 // 1 (shared by the batch) or B
    long long pse_s;
    const unsigned char* amask;// [B, amask_s] bytes, non zero = masked out, or null
    long long amask_s;
    // int8 output (…Us2.h:4016-4111): round(out * quant_scale2 + quant_offset2)
    const void* quant_scale2;  // [1] or [Hq * D], f32 or bf16, or null
    const void* quant_offset2; // same shape and dtype as quant_scale2, or null
    int quant2_dtype;          // kF32 | kBF16
    int quant2_per_channel;    // 1: [Hq * D]
    // Shared prefix (…Us2.h:1418-1439, 2263-2266): prefix_len tokens whose K / V rows every output row attends in front
    // of its gathered pages, held once in tensors of their own with the cache's token stride — BSH [1, S, Hkv * D] or
    // BNSD [1, Hkv, S, D].  They travel as leading page-list entries with phys = -1 - j (j-th run of block_size tokens).
    const void* prefix_k;      // or null
    const void* prefix_v;
    long long prefix_head_stride;   // bytes between two heads' rows of one token (BSH) / between two heads (BNSD)
    int prefix_len;            // attended prefix tokens
    int prefix_pages;          // ceil(prefix_len / block_size) leading list entries
    // softmax log-sum-exp of every output row (natural log, …Us2.h:3130-3159; -3.4e38 for a row without attended
    // tokens, :1289-1300), or null
    float* lse;                // [B, Hq]
    // mode 0
    // 1: block_position holds LOGICAL TOKEN indices (sentinel terminated; token t lives in row t % block_size of page
    // block_table[b][t / block_size]) and actual_seq_lengths counts tokens, the token-level variant sketched in the
    // reference's tests/ifa_v5_case_tkp.py:53-54; 0: logical page indices
    int token_mode;
    const int32_t* block_position; // [B, rows, P]
    int pos_rows;                  // Hq or Hkv
    const long long* act_seq;      // device [B] or null
    long long act_seq_uniform;     // >= 0: one length for every batch; < 0 and act_seq null: whole list
    // mode 1
    const void* cent;              // [Hkv, C, D] in q dtype
    const int32_t* block_ids;      // [Hkv, kv_page_len]
    const int32_t* seq_len;        // [B] total_seq_len
    int C, Cp, k, kv_page_len, sink_recent, cap_pages;
    int32_t* page_position;        // [B, Hq, P] out (may be null)
    int32_t* page_position_length; // [B, Hq, 8] out (may be null)
    unsigned long long* max_page_position_length; // [B, 8] out (may be null)
    int P;                         // slots per row (both modes)
    int rows_per_pass;             // q heads of a group handled per pass
    int max_entries;               // capacity of the per-workgroup page list
    unsigned* counters;            // start of the workspace: 8 work counters, [8] = workgroups that have left the kernel
    // The part of the workspace that every launch expects zeroed (counters, arrival words, boards): zero_words 8-byte
    // words from `counters` on.  The last workgroup to leave a launch zeroes it again, so a workspace that nobody
    // else writes needs the zeroing kernel only before its first launch.
    unsigned long long zero_words;
    // per-batch maximum of the fused op (cent_select.cpp:124-152) without pre-zeroed output: every item adds its
    // longest list to batch_max[b] and its row count to batch_arrive[b]; the item that completes the Hq rows of a
    // batch stores max_page_position_length[b, 0..7]
    unsigned* batch_arrive;        // [B] (workspace, zeroed)
    unsigned long long* batch_max; // [B] (workspace, zeroed)
    long long grid_cap;            // resident workgroups of the selected kernel on the planning device (phase 1); 0: ask at launch
    // 0: a workgroup claims items of its own XCD's class (item % 8) as long as there are any, so the kv heads an XCD
    // works on stay the same (centroids in its L2); 1: every claim starts one class further, which mixes heads whose
    // page lists are systematically longer or shorter over all XCDs
    int class_step;
    // 1: the streamer waves of a workgroup claim the pages of an item one by one from a counter in LDS (whoever is
    // free takes the next page); 0: wave w takes pages w, w + 4, ... which makes attention_out bit-reproducible
    int dyn_pages;
    // > 1: every workgroup owns exactly pre_items items (its own index, then one grid further each time), all of one kv
    // head: their rows are scored in ONE pass over the head's centroids before the first item is streamed
    int pre_items;
    // with pre_items > 1 — 1: the page lists of all those items are built before the first is streamed as well (one wave
    // per row); 0: only the scores are taken ahead, top-k and page selection of items 2.. stay with the selector waves
    int pre_lists;
    // > 1 (fused op, one q head per pass): the queue hands out super-items of up to sup_items consecutive sequences of
    // one kv head; the selector wave scores their rows in ONE pass over the head's centroids while the super-item
    // before is streamed (the first one: by all waves, before anything is streamed)
    int sup_items;
    // The super-items of a launch come in up to four phases of decreasing size: phase i holds sup_n[i] super-items of
    // sup_j[i] sequences for every kv head, numbered head-minor (super-item % heads = kv head) and handed out in that
    // order, so every workgroup gets large ones first and the small ones even the launch out at its end.
    int sup_j[4], sup_n[4];
    // page-list split for small batches: n_split workgroups per item (1: off)
    int n_split;
    // sharing of every workgroup's last item (n_split == 1 only): boards and their data areas in the workspace
    int tail_on;
    int tail_chunks;               // chunks a shared item is cut into (<= 64)
    void* tail_boards;             // [2 * #CU] 16-byte boards, zeroed before every launch
    char* tail_data;               // [2 * #CU] data areas of tail_stride bytes
    long long tail_stride;
    // shared first scoring (fused op): groups of workgroups score their first items' rows in one centroid pass
    int pg_on;
    unsigned* pg_done;             // [groups] arrival counters (workspace), zeroed before every launch
    uint32_t* pg_keys;             // [groups][16][Cp] score keys
    unsigned* split_done;          // [items] arrival counters (workspace), zeroed before every launch
    float* split_partials;         // [items * n_split][rows_per_pass][D + 2] un-normalised states (workspace)
    unsigned long long* debug;     // diagnostic builds only: [grid][16 items][16 stamps], else null
    // 1: the plain one-q-head kernel runs as two workgroups of four streamer waves per CU instead of one of eight (twice
    // the selector waves per CU; taken for fused launches with many short items per workgroup)
    int narrow;
    // waves of a workgroup that share the centroid pass of the first item (each needs an LDS tile of its own; the
    // other waves wait): all of them for head dims up to 128, fewer for longer rows (VALU form)
    int score_waves;
    // LDS carve (bytes from the dynamic base)
    int off_keys, off_cbits, off_pbits, off_list, off_comb, off_qst, off_hdr, off_vst, off_stage;
    int off_carry;            // token-level lists streamed in windows: two merged states [rows_per_pass][D + 2]
    int k_lds;                // matrix-core streamer: K is fetched in whole rows (every request a full cache line) and
                              // turned into operand order through a wave-private LDS image, as V is
    int vst_stride;           // bytes of tile images per streamer wave at off_vst
    size_t lds_bytes;
};

hipError_t launch_compute_cent(const CentParams& p, hipStream_t s);
hipError_t launch_cent_select(const CentParams& p, hipStream_t s);
hipError_t launch_select_position(const SelectParams& p, hipStream_t s);
hipError_t launch_assign(const AssignParams& p, hipStream_t s);
// LDS of the assign kernel for rows_per_pass rows: keys, the A tile, the slot list, the centroid tiles of its 4 waves.
size_t assign_lds_bytes(int rows_per_pass, int Cp, int D);
// cap_out != nullptr: launch nothing, report how many workgroups of the kernel that would run can be resident.
hipError_t launch_attention(const AttnParams& p, hipStream_t s, long long* cap_out = nullptr);
// Enqueue the zeroing of up to two 8-byte aligned ranges (either may be null / empty) as one small kernel.
hipError_t launch_zero(void* a, size_t a_bytes, void* b, size_t b_bytes, hipStream_t s);
// Enqueue dst[0..n) = host[0..n): the values travel as kernel arguments (copied when the launch is enqueued or captured),
// so the host array may die as soon as the call returns and nothing is allocated or waited for.
hipError_t launch_store_lengths(long long* dst, const long long* host, int n, hipStream_t s);
// Library-owned scratch of the current device (a file-scope __device__ array, zero when the library is loaded) for the
// entry point that takes no workspace: its device address and size.  Looked up once per device.
constexpr size_t kStaticScratchBytes = (size_t)32 << 20;
// Its layout: [0, kScratchZeroBytes) the zeroed part (counters, arrival words, boards) of launches through the entry
// point that takes no workspace; then kScratchSlots slots of kScratchSlotBytes, the zeroed parts of launches that bring
// a workspace, one slot per (device, stream); from kScratchDataOff on the data areas of the launches without workspace.
constexpr size_t kScratchZeroBytes = (size_t)1 << 20;
constexpr size_t kScratchSlotsOff = kScratchZeroBytes;
constexpr size_t kScratchSlotBytes = (size_t)64 << 10;
constexpr int kScratchSlots = 16;
constexpr size_t kScratchDataOff = kScratchSlotsOff + kScratchSlots * kScratchSlotBytes;
constexpr int kMaxScratchDevices = 64;   // devices the per-device tables of the library are sized for
hipError_t static_scratch(void** ptr, int* device);
// Kernels this library has enqueued (or recorded into a graph) since it was loaded, all threads and devices: lets a
// caller check "one kernel per call" without a profiler.
void count_launch();
unsigned long long launches_so_far();
// Compute units of the current device (0 when no device can be queried).
int device_cu_count();
// Number of q heads of a group one pass over the K/V pages serves, and whether that pass runs on the MFMA streamer
// (which stages V tiles in LDS: 4 waves x 32 rows x (2 D + 32) bytes at off_vst).
int attention_rows_per_pass(int G, int kv_dtype, int D);
bool attention_uses_mfma(int rows_per_pass, int kv_dtype, int D);
// Streamer waves per workgroup of the two kernel forms (sizes the state and V-tile areas in LDS).
int attention_stream_waves(bool mfma, int kv_dtype, bool narrow);
// LDS tiles through which the centroid pass turns memory order into matrix-core operand order: bytes at off_stage of
// the attention kernels, and at the end of the dynamic LDS of the cent kernels.
// bytes of LDS per streamer wave of the matrix-core form: the V image of a 32-token tile (16-bit elements, rows padded by
// 32 B).  With k_lds the K image of the tile (elements as stored, rows padded by 16 B: never larger) lies in the same
// bytes: a wave has read its K operands back before it writes the V rows, and LDS executes a wave's operations in order.
constexpr int mfma_tile_area(int D) { return 32 * (D * 2 + 32); }
bool attention_team(const AttnParams& p);   // the plan's kernel runs a team of selector waves
// sup: such a plan
int attention_stage_bytes(bool mfma, int kv_dtype, int D, bool narrow, bool sup);
int attention_score_waves(bool mfma, int kv_dtype, int D, bool narrow, bool sup);   // AttnParams::score_waves of a plan
int cent_score_waves(int D);     // waves that share the pass of the standalone scoring kernels
int cent_stage_bytes(int D);      // their tiles
int assign_stage_bytes(int D);    // 4 tiles

}  // namespace spfa
