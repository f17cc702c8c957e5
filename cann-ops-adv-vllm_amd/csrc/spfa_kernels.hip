// gfx950 (MI355X / CDNA4) kernels for the sparse paged fusion attention chain:
//   ComputeCent -> CentSelect / SelectPosition -> SparsePagedAttention, and the fused op.
//
// Design (wave64, 4 waves per workgroup, one workgroup per (batch, kv head)):
//   * Q.Cent^T   : v_mfma_f32_16x16x32_{f16,bf16}; A = the group's q heads (zero padded to 16
//                  rows), B = 16 centroid rows loaded 16 B per lane straight from global/L2.
//                  fp16 x fp16 products are exact in fp32, accumulation is fp32 like the
//                  reference's MulAddDst (compute_cent.cpp:279-340).
//   * top-k      : one wave per q head.  Scores become order-preserving uint32 keys in LDS; the
//                  k-th largest key is found by a 32-step bitwise bisection whose counts are
//                  wave ballots, ties are broken by ascending cluster id.  The k ids become a
//                  C-bit bitmap in LDS (cent_select / fused) or are sorted descending (compute_cent).
//   * page select: 64 logical pages per step: cluster id = block_ids[kv][block_table[b][i]],
//                  membership = one bitmap probe, ballot + prefix popcount gives the output slot,
//                  so page_position is written in ascending order with coalesced int32 stores.
//   * attention  : HBM-bound (2*G flop per KV byte).  K/V rows go from HBM straight to VGPRs with
//                  16 B per lane (a 256 B fp16 head row = 16 lanes), every 16-lane token group keeps
//                  its own online-softmax state (m, l, o) so the page loop has no cross-lane traffic
//                  beyond a 16-lane DPP reduction of the dot product; states are merged once at the
//                  end (shuffles inside a wave, LDS across the 4 waves).  A GQA group walks the
//                  UNION of its heads' page lists once; a per-page head mask gates each head.
//   * int8 KV    : dequant is fused into the load path: q' = q * scale_k, bytes are converted with
//                  v_cvt_f32_ubyte after flipping the sign bit (u = x + 128), the -128 bias is
//                  folded into the accumulators, out = (P.V / l + offset_v) * scale_v
//                  (sparse_paged_fusion_attention_split_Bbn2s2_Us2.h:2647,2651,4160-4174).
#include "spfa_params.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spfa {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ unsigned long long lanes_below(int lane) { return (1ull << lane) - 1ull; }

// LDS traffic of ONE wave is in order; this only stops the compiler from caching or moving it.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// fp32 -> uint32 whose unsigned order equals the float order (-0 == +0, +NaN largest).
__device__ __forceinline__ uint32_t score_key(float x) {
    x = x + 0.0f;
    uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int CTRL>
__device__ __forceinline__ float dpp(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// Sum over the LPT consecutive lanes that share one token; every lane of the group gets the sum.
template <int LPT>
__device__ __forceinline__ float group_allsum(float x) {
    if constexpr (LPT >= 32) x += __shfl_xor(x, 16);
    if constexpr (LPT >= 16) { x += dpp<0x128>(x); x += dpp<0x124>(x); }   // row_ror 8, 4
    else if constexpr (LPT == 8) x += dpp<0x141>(x);                       // row_half_mirror
    x += dpp<0x4E>(x);                                                     // quad_perm [2,3,0,1]
    x += dpp<0xB1>(x);                                                     // quad_perm [1,0,3,2]
    return x;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float f16_bits(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ float load_qk_elem(const void* p, int dtype, long long i) {
    uint16_t h = ((const uint16_t*)p)[i];
    return dtype == kBF16 ? __uint_as_float((uint32_t)h << 16) : f16_bits(h);
}

// ------------------------------------------------------------------------------------------------
// Q.Cent^T for R (<= 16) q heads of one (batch, kv head) -> order keys in LDS [R][Cp]
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void score_rows(const void* q, const void* cent, int dtype, long long q_row0,
                                           int n2, int R, int D, int C, int Cp, uint32_t* keys) {
    const int lane = lane_id(), wave = wave_id();
    const int col = lane & 15;   // A: q row of the tile, B: centroid of the tile
    const int kq = lane >> 4;    // which 8-wide slice of the 32-deep k step
    const int KS = D >> 5;
    uint4 qa[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        qa[kk] = make_uint4(0, 0, 0, 0);
        if (kk < KS && col < R)
            qa[kk] = *(const uint4*)((const char*)q + ((q_row0 + col) * D + kk * 32 + kq * 8) * 2);
    }
    const char* cbase = (const char*)cent + (long long)n2 * C * D * 2;
    const int ntiles = Cp >> 4;

    for (int t = wave; t < ntiles; t += 4) {
        const int c = t * 16 + col;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (kk < KS) {
                uint4 bv = make_uint4(0, 0, 0, 0);
                if (c < C) bv = *(const uint4*)(cbase + ((long long)c * D + kk * 32 + kq * 8) * 2);
                if (dtype == kBF16)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, qa[kk]),
                                                                  __builtin_bit_cast(bf16x8, bv), acc, 0, 0, 0);
                else
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, qa[kk]),
                                                                 __builtin_bit_cast(half8, bv), acc, 0, 0, 0);
            }
        }
        // D fragment: row = (lane >> 4) * 4 + r (q head), col = lane & 15 (centroid)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qr = kq * 4 + r;
            if (qr < R) keys[qr * Cp + c] = (c < C) ? score_key(acc[r]) : 0u;
        }
    }
}

// k-th largest key of krow[0..Cp) (Cp % 64 == 0); executed by one full wave.
__device__ __forceinline__ uint32_t kth_largest_key(const uint32_t* krow, int Cp, int k, int lane) {
    uint32_t T = 0;
    if (Cp <= 2048) {
        uint32_t kr[32];
        const int nj = Cp >> 6;
#pragma unroll
        for (int j = 0; j < 32; ++j) kr[j] = (j < nj) ? krow[j * 64 + lane] : 0u;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = T | (1u << bit);
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) cnt += __popcll(__ballot(kr[j] >= cand));
            if (cnt >= k) T = cand;
        }
    } else {
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = T | (1u << bit);
            int cnt = 0;
            for (int j = lane; j < Cp; j += 64) cnt += __popcll(__ballot(krow[j] >= cand));
            if (cnt >= k) T = cand;
        }
    }
    return T;
}

// Top-k of one row as a C-bit bitmap (uint32 words, Cp/32 of them); one full wave.
__device__ __forceinline__ void topk_bitmap(const uint32_t* krow, int Cp, int k, uint32_t* bits, int lane) {
    const uint32_t T = kth_largest_key(krow, Cp, k, lane);
    int n_gt = 0;
    for (int j = lane; j < Cp; j += 64) n_gt += __popcll(__ballot(krow[j] > T));
    int need = k - n_gt;   // ties at the threshold, lowest cluster ids first
    for (int j = lane; j < Cp; j += 64) {
        const uint32_t key = krow[j];
        const unsigned long long eq = __ballot(key == T);
        const bool take = (key > T) || (key == T && __popcll(eq & lanes_below(lane)) < need);
        const unsigned long long sel = __ballot(take);
        need -= min(need, (int)__popcll(eq));
        if (lane == 0) {
            bits[(j >> 5)] = (uint32_t)sel;
            bits[(j >> 5) + 1] = (uint32_t)(sel >> 32);
        }
    }
}

// Top-k of one row, sorted descending (ties: ascending id) -> out[0..k); one full wave.
// sortbuf: LDS, npow2 (>= k, power of two) uint64 entries owned by this wave.
__device__ __forceinline__ void topk_sorted(const uint32_t* krow, int Cp, int k, int npow2,
                                            unsigned long long* sortbuf, int32_t* out, int lane) {
    const uint32_t T = kth_largest_key(krow, Cp, k, lane);
    int n_gt = 0;
    for (int j = lane; j < Cp; j += 64) n_gt += __popcll(__ballot(krow[j] > T));
    int need = k - n_gt;
    int base = 0;
    for (int j = lane; j < Cp; j += 64) {
        const uint32_t key = krow[j];
        const unsigned long long eq = __ballot(key == T);
        const bool take = (key > T) || (key == T && __popcll(eq & lanes_below(lane)) < need);
        const unsigned long long sel = __ballot(take);
        need -= min(need, (int)__popcll(eq));
        if (take) sortbuf[base + __popcll(sel & lanes_below(lane))] =
            ((unsigned long long)key << 32) | (uint32_t)(~(uint32_t)j);
        base += __popcll(sel);
    }
    for (int i = k + lane; i < npow2; i += 64) sortbuf[i] = 0ull;
    wave_lds_sync();
    for (int size = 2; size <= npow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = lane; t < (npow2 >> 1); t += 64) {
                const int i = 2 * t - (t & (stride - 1));
                const int j = i + stride;
                const bool first_half = (i & size) == 0;
                const unsigned long long a = sortbuf[i], b = sortbuf[j];
                if ((a < b) == first_half) { sortbuf[i] = b; sortbuf[j] = a; }
            }
            wave_lds_sync();
        }
    }
    for (int i = lane; i < k; i += 64) out[i] = (int32_t)(~(uint32_t)sortbuf[i]);
}

// Page selection for one q head by one full wave.
//   cbits : the head's top-k cluster bitmap;  pbits (optional): 64-bit words, bit i of word w set
//   when logical page 64*w + i is attended (i.e. selected and inside the first `limit` hits).
// Returns the stored length min(count, limit).
__device__ __forceinline__ int select_pages(const uint32_t* cbits, int C, const int32_t* bt_row,
                                            const int32_t* ids_row, int kv_page_len, int page_len,
                                            int P, bool sink_recent, bool cap_pages, int32_t* pos_row,
                                            unsigned long long* pbits, int n_pwords, int lane) {
    int bt0 = 0, btm1 = 0, btm2 = 0;
    if (sink_recent && page_len > 0) {   // cent_select.cpp:350-362: compare block_table VALUES
        bt0 = bt_row[0];
        btm1 = bt_row[page_len - 1];
        btm2 = bt_row[page_len >= 2 ? page_len - 2 : 0];
    }
    const int limit = cap_pages ? min(P, page_len / 8) : P;   // cent_select.cpp:218,410-414
    int count = 0;
    for (int base = 0; base < page_len; base += 64) {
        const int i = base + lane;
        const bool in = i < page_len;
        int raw = in ? bt_row[i] : 0;
        const int pg = min(max(raw, 0), kv_page_len - 1);   // tests/utils.py:62 clamps
        const int cid = in ? ids_row[pg] : -1;
        bool hit = in && cid >= 0 && cid < C && ((cbits[cid >> 5] >> (cid & 31)) & 1u);
        if (sink_recent) hit = hit || (in && (raw == bt0 || raw == btm1 || raw == btm2));
        const unsigned long long hb = __ballot(hit);
        const int slot = count + __popcll(hb & lanes_below(lane));
        if (hit && slot < P && pos_row) pos_row[slot] = i;
        if (pbits) {
            const unsigned long long ab = __ballot(hit && slot < limit);
            if (lane == 0) pbits[base >> 6] = ab;
        }
        count += __popcll(hb);
    }
    if (pbits)
        for (int w = ((page_len + 63) >> 6) + lane; w < n_pwords; w += 64) pbits[w] = 0ull;
    if (pos_row)
        for (int s = min(count, P) + lane; s < P; s += 64) pos_row[s] = kSentinel;
    return min(count, limit);
}

// ------------------------------------------------------------------------------------------------
// compute_cent / cent_select kernel
// ------------------------------------------------------------------------------------------------
template <int MODE>   // 0: compute_cent, 1: cent_select
__global__ __launch_bounds__(kThreads) void cent_kernel(const CentParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id(), wave = wave_id();
    const int n2 = blockIdx.x % p.Hkv;   // blocks b and b+8 share an XCD: one kv head's centroids
    const int b = blockIdx.x / p.Hkv;    // stay in that XCD's L2
    const int RG = p.rows_per_pass;
    uint32_t* keys = (uint32_t*)smem;                                  // [RG][Cp]
    char* aux = smem + (size_t)RG * p.Cp * 4;
    int page_len = 0;
    if (MODE == 1) page_len = (p.seq_len[b] + p.block_size - 1) / p.block_size;   // cent_select.cpp:217
    page_len = max(0, min(page_len, p.max_page));
    int len_max = 0;
    for (int g0 = 0; g0 < p.G; g0 += RG) {
        const int R = min(RG, p.G - g0);
        const long long row0 = (long long)b * p.Hq + (long long)n2 * p.G + g0;
        score_rows(p.q, p.cent, p.qk_dtype, row0, n2, R, p.D, p.C, p.Cp, keys);
        __syncthreads();
        for (int r = wave; r < R; r += 4) {
            const long long row = row0 + r;
            if (MODE == 0) {
                int npow2 = 64;
                while (npow2 < p.k) npow2 <<= 1;
                unsigned long long* sortbuf = (unsigned long long*)aux + (size_t)wave * npow2;
                topk_sorted(keys + r * p.Cp, p.Cp, p.k, npow2, sortbuf, p.indices + row * p.k, lane);
            } else {
                uint32_t* cbits = (uint32_t*)aux + (size_t)wave * (p.Cp >> 5);
                topk_bitmap(keys + r * p.Cp, p.Cp, p.k, cbits, lane);
                wave_lds_sync();
                const int len = select_pages(cbits, p.C, p.block_table + (long long)b * p.max_page,
                                             p.block_ids + (long long)n2 * p.kv_page_len, p.kv_page_len,
                                             page_len, p.P, p.sink_recent, p.cap_pages,
                                             p.page_position + row * p.P, nullptr, 0, lane);
                if (lane < kLenPad) p.page_position_length[row * kLenPad + lane] = len;   // cent_select.cpp:418
                len_max = max(len_max, len);
            }
        }
        __syncthreads();
    }
    if (MODE == 1 && lane < kLenPad && len_max > 0)   // cent_select.cpp:124-152: max_h(len) * page size
        atomicMax(p.max_page_position_length + (long long)b * kLenPad + lane,
                  (unsigned long long)len_max * (unsigned long long)p.block_size);
}

// ------------------------------------------------------------------------------------------------
// select_position kernel: indices are given (any int32 values, repeats allowed)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void select_position_kernel(const SelectParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id(), wave = wave_id();
    const long long row = (long long)blockIdx.x * 4 + wave;   // (b, q head)
    if (row >= (long long)p.B * p.Hq) return;
    const int b = (int)(row / p.Hq), h = (int)(row % p.Hq), n2 = h / p.G;
    int32_t* idx = (int32_t*)smem + (size_t)wave * p.k;
    for (int j = lane; j < p.k; j += 64) idx[j] = p.indices[row * p.k + j];
    wave_lds_sync();
    int page_len = (p.seq_len[b] + p.block_size - 1) / p.block_size;   // select_position.cpp:112
    page_len = max(0, min(page_len, p.max_page));
    const int32_t* bt_row = p.block_table + (long long)b * p.max_page;
    const int32_t* ids_row = p.block_ids + (long long)n2 * p.kv_page_len;
    int32_t* pos_row = p.page_position + row * p.P;
    int32_t* gat_row = p.block_table_gather + row * p.max_page;
    int count = 0;
    for (int base = 0; base < p.max_page; base += 64) {
        const int i = base + lane;
        const bool in = i < page_len;
        int cid = 0;
        bool hit = false;
        if (in) {
            const int pg = min(max(bt_row[i], 0), p.kv_page_len - 1);
            cid = ids_row[pg];                                          // select_position.cpp:126,136
            for (int j = 0; j < p.k; ++j) hit = hit || (idx[j] == cid); // :158-164
        }
        if (i < p.max_page) gat_row[i] = in ? cid : 0;                  // :143 (stride max_page)
        const unsigned long long hb = __ballot(hit);
        const int slot = count + __popcll(hb & lanes_below(lane));
        if (hit && slot < p.P) pos_row[slot] = i;                       // :174,177 ascending compaction
        count += __popcll(hb);
    }
    const int len = min(count, p.P);
    for (int s = len + lane; s < p.P; s += 64) pos_row[s] = kSentinel;  // :74,186-188
    if (lane < kLenPad) p.page_position_length[row * kLenPad + lane] = len;   // :78,191-193
}

// ------------------------------------------------------------------------------------------------
// attention
// ------------------------------------------------------------------------------------------------
struct PageEntry {            // 16 bytes in LDS
    int phys;                 // physical block
    int valid;                // attended tokens of the page
    unsigned long long mask;  // bit g: head g of the pass attends this page
};

template <typename KVT> struct KvTraits;
template <> struct KvTraits<_Float16> { static constexpr int EPL = 8; };
template <> struct KvTraits<__bf16>   { static constexpr int EPL = 8; };
template <> struct KvTraits<int8_t>   { static constexpr int EPL = 16; };

template <typename KVT>
__device__ __forceinline__ void unpack16(const uint4& v, float* f) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    if constexpr (sizeof(KVT) == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t u = w[i] ^ 0x80808080u;   // x + 128 as unsigned bytes
            f[4 * i + 0] = (float)(u & 0xffu);
            f[4 * i + 1] = (float)((u >> 8) & 0xffu);
            f[4 * i + 2] = (float)((u >> 16) & 0xffu);
            f[4 * i + 3] = (float)(u >> 24);
        }
    } else if constexpr (__is_same(KVT, __bf16)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[2 * i] = bf16_lo(w[i]); f[2 * i + 1] = bf16_hi(w[i]); }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = f16_bits((uint16_t)(w[i] & 0xffffu));
            f[2 * i + 1] = f16_bits((uint16_t)(w[i] >> 16));
        }
    }
}

// Merge two online-softmax states (base-2 exponents).
__device__ __forceinline__ void merge_state(float& m, float& l, float* o, int n, float m2, float l2, const float* o2) {
    const float mt = fmaxf(m, m2);
    const float a = fast_exp2(m - mt), c = fast_exp2(m2 - mt);
    l = l * a + l2 * c;
    for (int e = 0; e < n; ++e) o[e] = o[e] * a + o2[e] * c;
    m = mt;
}

template <typename KVT, int NQ, int LPT>
__global__ __launch_bounds__(kThreads) void attn_kernel(const AttnParams p) {
    constexpr int EPL = KvTraits<KVT>::EPL;     // elements per 16-byte lane load
    constexpr int TPL = 64 / LPT;               // tokens per wave load
    constexpr int LG = (NQ * EPL <= 16) ? 8 : ((NQ * EPL <= 32) ? 4 : 2);   // loads per step and cache
    constexpr bool QUANT = sizeof(KVT) == 1;
    constexpr float kNegBig = -1.0e30f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id(), wave = wave_id();
    const int sub = lane % LPT;                 // 16-byte slice of the head row
    const int grp = lane / LPT;                 // token within a wave load
    const int D = p.D;

    int b, n2, head0, G_eff;
    if (p.mode == 0 && p.pos_rows == p.Hq && p.G > 1) {   // every q head has its own list
        b = blockIdx.x / p.Hq;
        head0 = blockIdx.x % p.Hq;
        n2 = head0 / p.G;
        G_eff = 1;
    } else {
        n2 = blockIdx.x % p.Hkv;
        b = blockIdx.x / p.Hkv;
        head0 = n2 * p.G;
        G_eff = p.G;
    }
    PageEntry* list = (PageEntry*)smem;                          // aliases the score keys
    uint32_t* keys = (uint32_t*)smem;
    uint32_t* cbits_all = (uint32_t*)(smem + p.off_cbits);       // [4 waves][Cp/32]
    unsigned long long* pbits_all = (unsigned long long*)(smem + p.off_pbits);   // [NQ][n_pwords]
    int* misc = (int*)(smem + p.off_misc);                       // [0] = list length
    const int n_pwords = (p.max_page + 63) >> 6;
    const int32_t* bt_row = p.block_table + (long long)b * p.max_page;

    int page_len = 0, seq = 0, len_max = 0;
    if (p.mode == 1) {
        seq = p.seq_len[b];
        page_len = max(0, min((seq + p.block_size - 1) / p.block_size, p.max_page));
    }
    const float qscale = p.scale * 1.4426950408889634f;   // scores in base-2 exponent units

    for (int g0 = 0; g0 < G_eff; g0 += NQ) {
        const int R = min(NQ, G_eff - g0);
        const long long row0 = (long long)b * p.Hq + head0 + g0;
        // ---------------- phase A: page list of this pass into LDS ----------------
        if (p.mode == 1) {
            score_rows(p.q, p.cent, p.q_dtype, row0, n2, R, D, p.C, p.Cp, keys);
            __syncthreads();
            for (int r = wave; r < R; r += 4) {
                uint32_t* cbits = cbits_all + (size_t)wave * (p.Cp >> 5);
                topk_bitmap(keys + r * p.Cp, p.Cp, p.k, cbits, lane);
                wave_lds_sync();
                const long long row = row0 + r;
                const int len = select_pages(cbits, p.C, bt_row, p.block_ids + (long long)n2 * p.kv_page_len,
                                             p.kv_page_len, page_len, p.P, p.sink_recent, p.cap_pages,
                                             p.page_position ? p.page_position + row * p.P : nullptr,
                                             pbits_all + (size_t)r * n_pwords, n_pwords, lane);
                if (p.page_position_length && lane < kLenPad) p.page_position_length[row * kLenPad + lane] = len;
                len_max = max(len_max, len);
            }
            __syncthreads();   // keys are dead from here on: the list may overwrite them
            if (wave == 0) {
                int nU = 0;
                for (int w = 0; w < ((page_len + 63) >> 6); ++w) {
                    unsigned long long u = 0, mine = 0;
                    for (int r = 0; r < R; ++r) {
                        const unsigned long long x = pbits_all[(size_t)r * n_pwords + w];
                        u |= x;
                        mine |= ((x >> lane) & 1ull) << r;
                    }
                    if ((u >> lane) & 1ull) {
                        const int i = w * 64 + lane;
                        const int e = nU + __popcll(u & lanes_below(lane));
                        if (e < p.max_entries) {
                            PageEntry pe;
                            pe.phys = min(max(bt_row[i], 0), p.n_blocks - 1);
                            pe.valid = min(p.block_size, seq - i * p.block_size);
                            pe.mask = mine;
                            list[e] = pe;
                        }
                    }
                    nU += __popcll(u);
                }
                if (lane == 0) misc[0] = min(nU, p.max_entries);
            }
        } else {
            // list = leading non-sentinel slots of the row, bounded by actual_seq_lengths
            // (…Us2.h:256-294; the kernel stops at the first 0x7fffffff, :308-311)
            if (wave == 0) {
                const int prow = (p.pos_rows == p.Hq) ? (head0 + g0) : n2;
                const int32_t* pos = p.block_position + ((long long)b * p.pos_rows + prow) * p.P;
                long long n_tok = p.act_seq ? p.act_seq[b] : (p.act_seq_uniform >= 0 ? p.act_seq_uniform : (long long)p.P * p.block_size);
                n_tok = max(0ll, min(n_tok, (long long)p.P * p.block_size));
                const int n_slots = (int)((n_tok + p.block_size - 1) / p.block_size);
                int nU = 0;
                bool open = true;
                for (int base = 0; base < n_slots && open; base += 64) {
                    const int j = base + lane;
                    const int pg = (j < n_slots) ? pos[j] : kSentinel;
                    const bool stop = (pg == kSentinel) || pg < 0 || pg >= p.max_page;
                    const unsigned long long sb = __ballot(stop);
                    const int first = sb ? (__ffsll((long long)sb) - 1) : 64;
                    if (lane < first && nU + lane < p.max_entries) {
                        PageEntry pe;
                        pe.phys = min(max(bt_row[pg], 0), p.n_blocks - 1);
                        pe.valid = (int)min((long long)p.block_size, n_tok - (long long)j * p.block_size);
                        pe.mask = ~0ull;
                        list[nU + lane] = pe;
                    }
                    nU += first;
                    open = (first == 64);
                }
                if (lane == 0) misc[0] = min(nU, p.max_entries);
            }
        }
        __syncthreads();
        const int nU = misc[0];

        // ---------------- phase B: stream the pages ----------------
        float qf[NQ][EPL];
        float qbias[NQ];   // int8: 128 * sum of this lane's q' (u = x + 128)
#pragma unroll
        for (int g = 0; g < NQ; ++g) {
            qbias[g] = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                float v = 0.f;
                if (g < R) {
                    v = load_qk_elem(p.q, p.q_dtype, (row0 + g) * D + sub * EPL + e) * qscale;
                    if (QUANT) v *= load_qk_elem(p.antiq_scale, p.q_dtype, (long long)n2 * D + sub * EPL + e);
                }
                qf[g][e] = v;
                qbias[g] += 128.f * v;
            }
        }
        float o[NQ][EPL], m[NQ], l[NQ];
#pragma unroll
        for (int g = 0; g < NQ; ++g) {
            m[g] = kNegBig;
            l[g] = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) o[g][e] = 0.f;
        }
        const unsigned lane_off = (unsigned)(sub * 16);
        for (int n = wave; n < nU; n += 4) {
            const int phys = __builtin_amdgcn_readfirstlane(list[n].phys);
            const int valid = __builtin_amdgcn_readfirstlane(list[n].valid);
            const unsigned mlo = __builtin_amdgcn_readfirstlane((unsigned)list[n].mask);
            const char* kp = (const char*)p.kcache + (long long)phys * p.blk_stride + (long long)n2 * p.head_stride;
            const char* vp = (const char*)p.vcache + (long long)phys * p.blk_stride + (long long)n2 * p.head_stride;
            for (int t0 = 0; t0 < valid; t0 += TPL * LG) {
                uint4 kreg[LG], vreg[LG];
#pragma unroll
                for (int i = 0; i < LG; ++i) {
                    const int tok = min(t0 + i * TPL + grp, p.block_size - 1);
                    kreg[i] = *(const uint4*)(kp + (unsigned)(tok * (int)p.tok_stride) + lane_off);
                }
#pragma unroll
                for (int i = 0; i < LG; ++i) {
                    const int tok = min(t0 + i * TPL + grp, p.block_size - 1);
                    vreg[i] = *(const uint4*)(vp + (unsigned)(tok * (int)p.tok_stride) + lane_off);
                }
                float s[NQ][LG];
#pragma unroll
                for (int i = 0; i < LG; ++i) {
                    float kf[EPL];
                    unpack16<KVT>(kreg[i], kf);
                    const bool live = (t0 + i * TPL + grp) < valid;
#pragma unroll
                    for (int g = 0; g < NQ; ++g) {
                        float acc = QUANT ? -qbias[g] : 0.f;
#pragma unroll
                        for (int e = 0; e < EPL; ++e) acc = fmaf(qf[g][e], kf[e], acc);
                        acc = group_allsum<LPT>(acc);
                        s[g][i] = live ? acc : -INFINITY;
                    }
                }
#pragma unroll
                for (int g = 0; g < NQ; ++g) {
                    if (!((mlo >> g) & 1u)) {   // wave-uniform: this head does not attend the page
#pragma unroll
                        for (int i = 0; i < LG; ++i) s[g][i] = 0.f;
                        continue;
                    }
                    float cmax = s[g][0];
#pragma unroll
                    for (int i = 1; i < LG; ++i) cmax = fmaxf(cmax, s[g][i]);
                    const float mn = fmaxf(m[g], cmax);
                    const float alpha = fast_exp2(m[g] - mn);
                    m[g] = mn;
                    float psum = 0.f;
#pragma unroll
                    for (int i = 0; i < LG; ++i) { s[g][i] = fast_exp2(s[g][i] - mn); psum += s[g][i]; }
                    l[g] = l[g] * alpha + psum;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) o[g][e] *= alpha;
                }
#pragma unroll
                for (int i = 0; i < LG; ++i) {
                    float vf[EPL];
                    unpack16<KVT>(vreg[i], vf);
#pragma unroll
                    for (int g = 0; g < NQ; ++g) {
                        if ((mlo >> g) & 1u) {
#pragma unroll
                            for (int e = 0; e < EPL; ++e) o[g][e] = fmaf(s[g][i], vf[e], o[g][e]);
                        }
                    }
                }
            }
        }
        // ---------------- merge the 64/LPT token groups of the wave, then the 4 waves ----------------
#pragma unroll
        for (int g = 0; g < NQ; ++g) {
            if (QUANT) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) o[g][e] -= 128.f * l[g];   // undo u = x + 128
            }
#pragma unroll
            for (int step = LPT; step < 64; step <<= 1) {
                float o2[EPL];
                const float m2 = __shfl_xor(m[g], step), l2 = __shfl_xor(l[g], step);
#pragma unroll
                for (int e = 0; e < EPL; ++e) o2[e] = __shfl_xor(o[g][e], step);
                merge_state(m[g], l[g], o[g], EPL, m2, l2, o2);
            }
        }
        __syncthreads();   // every wave is done with the page list: reuse its LDS for the merge
        float* comb = (float*)smem;   // [4 waves][NQ][D + 2]
        const int cstride = D + 2;
#pragma unroll
        for (int g = 0; g < NQ; ++g) {
            float* dst = comb + ((size_t)wave * NQ + g) * cstride;
            if (grp == 0) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) dst[sub * EPL + e] = o[g][e];
                if (sub == 0) { dst[D] = m[g]; dst[D + 1] = l[g]; }
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < R * D; t += kThreads) {
            const int g = t / D, d = t % D;
            float mt = kNegBig;
            for (int w = 0; w < 4; ++w) mt = fmaxf(mt, comb[((size_t)w * NQ + g) * cstride + D]);
            float acc = 0.f, lt = 0.f;
            for (int w = 0; w < 4; ++w) {
                const float* src = comb + ((size_t)w * NQ + g) * cstride;
                const float c = fast_exp2(src[D] - mt);
                acc += src[d] * c;
                lt += src[D + 1] * c;
            }
            float r = lt > 0.f ? acc / lt : 0.f;
            if (QUANT && lt > 0.f) {   // (P.V / l + offset_v) * scale_v, …Us2.h:4160-4174
                const long long pi = (long long)p.Hkv * D + (long long)n2 * D + d;   // value params follow the key params
                if (p.antiq_offset) r += load_qk_elem(p.antiq_offset, p.q_dtype, pi);
                r *= load_qk_elem(p.antiq_scale, p.q_dtype, pi);
            }
            const long long oi = (row0 + g) * D + d;
            if (p.out_dtype == kF32) ((float*)p.out)[oi] = r;
            else if (p.out_dtype == kBF16) ((__bf16*)p.out)[oi] = (__bf16)r;   // round to nearest even (…Us2.h:3476-3477)
            else ((_Float16*)p.out)[oi] = (_Float16)r;
        }
        __syncthreads();
    }
    if (p.mode == 1 && p.max_page_position_length && lane < kLenPad && len_max > 0)
        atomicMax(p.max_page_position_length + (long long)b * kLenPad + lane,
                  (unsigned long long)len_max * (unsigned long long)p.block_size);
}

template <typename KVT, int NQ>
hipError_t launch_attn_lpt(const AttnParams& p, int lpt, dim3 grid, hipStream_t s) {
#define SPFA_LAUNCH(L)                                                                          \
    case L: {                                                                                   \
        auto kern = attn_kernel<KVT, NQ, L>;                                                    \
        if (p.lds_bytes > 48 * 1024) {                                                          \
            hipError_t e = hipFuncSetAttribute((const void*)kern,                               \
                                               hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                               (int)p.lds_bytes);                               \
            if (e != hipSuccess) return e;                                                      \
        }                                                                                       \
        hipLaunchKernelGGL(kern, grid, dim3(kThreads), p.lds_bytes, s, p);                      \
        return hipGetLastError();                                                               \
    }
    switch (lpt) {
        SPFA_LAUNCH(4)
        SPFA_LAUNCH(8)
        SPFA_LAUNCH(16)
        SPFA_LAUNCH(32)
    }
#undef SPFA_LAUNCH
    return hipErrorInvalidValue;
}

template <typename KVT>
hipError_t launch_attn_nq(const AttnParams& p, int nq, int lpt, dim3 grid, hipStream_t s) {
    switch (nq) {
        case 1: return launch_attn_lpt<KVT, 1>(p, lpt, grid, s);
        case 4: return launch_attn_lpt<KVT, 4>(p, lpt, grid, s);
        case 8: return launch_attn_lpt<KVT, 8>(p, lpt, grid, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace

int attention_rows_per_pass(int G, int kv_dtype) {
    if (G <= 1) return 1;
    return (G <= 4 || kv_dtype == kI8) ? 4 : 8;   // int8 rows cost 16 accumulators each
}

hipError_t launch_attention(const AttnParams& p, hipStream_t s) {
    const bool per_head = (p.mode == 0 && p.pos_rows == p.Hq && p.G > 1);
    const int nq = per_head ? 1 : p.rows_per_pass;
    const int esize = (p.kv_dtype == kI8) ? 1 : 2;
    const int lpt = p.D * esize / 16;
    const dim3 grid((unsigned)((long long)p.B * (per_head ? p.Hq : p.Hkv)));
    if (grid.x == 0) return hipSuccess;
    switch (p.kv_dtype) {
        case kF16: return launch_attn_nq<_Float16>(p, nq, lpt, grid, s);
        case kBF16: return launch_attn_nq<__bf16>(p, nq, lpt, grid, s);
        case kI8: return launch_attn_nq<int8_t>(p, nq, lpt, grid, s);
    }
    return hipErrorInvalidValue;
}

static hipError_t launch_cent(const CentParams& p, int mode, hipStream_t s) {
    const dim3 grid((unsigned)((long long)p.B * p.Hkv));
    if (grid.x == 0) return hipSuccess;
    const void* kern = mode == 0 ? (const void*)cent_kernel<0> : (const void*)cent_kernel<1>;
    if (p.lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        if (e != hipSuccess) return e;
    }
    if (mode == 0) hipLaunchKernelGGL(cent_kernel<0>, grid, dim3(kThreads), p.lds_bytes, s, p);
    else hipLaunchKernelGGL(cent_kernel<1>, grid, dim3(kThreads), p.lds_bytes, s, p);
    return hipGetLastError();
}
hipError_t launch_compute_cent(const CentParams& p, hipStream_t s) { return launch_cent(p, 0, s); }
hipError_t launch_cent_select(const CentParams& p, hipStream_t s) { return launch_cent(p, 1, s); }

hipError_t launch_select_position(const SelectParams& p, hipStream_t s) {
    const long long rows = (long long)p.B * p.Hq;
    if (rows == 0) return hipSuccess;
    const dim3 grid((unsigned)((rows + 3) / 4));
    hipLaunchKernelGGL(select_position_kernel, grid, dim3(kThreads), (size_t)4 * p.k * sizeof(int32_t), s, p);
    return hipGetLastError();
}

}  // namespace spfa
