// gfx950 (MI355X / CDNA4) kernels for the sparse paged fusion attention chain:
//   ComputeCent -> CentSelect / SelectPosition -> SparsePagedAttention, and the fused op.
//
// Building blocks (all wave64):
//   * Q.Cent^T   : v_mfma_f32_16x16x32_{f16,bf16}; A = the q heads of one kv head (zero padded to 16
//                  rows), B = 16 centroid rows loaded 16 B per lane straight from global/L2.
//                  fp16 x fp16 products are exact in fp32, accumulation is fp32 like the
//                  reference's MulAddDst (compute_cent.cpp:279-340).
//   * top-k      : one wave per q head.  Scores become order-preserving uint32 keys in LDS; the
//                  k-th largest key is found by an integer bisection whose full steps count with one
//                  compare per key and whose late steps run on at most 64 compacted candidates; ties
//                  are broken by ascending cluster id.  The k ids become a C-bit bitmap in LDS
//                  (cent_select / fused) or are sorted descending (compute_cent).
//   * page select: 256 logical pages per step: cluster id = block_ids[kv][block_table[b][i]],
//                  membership = one bitmap probe, ballot + prefix popcount gives the output slot,
//                  so page_position is written in ascending order with coalesced int32 stores.
//   * attention  : HBM-bound (2*G flop per KV byte).  A persistent grid of workgroups pulls (batch,
//                  kv head) items from a queue; selector wave(s) prepare the next item's page list
//                  while four streamer waves read the current item's K/V pages.  One q head per
//                  item: K/V rows go from HBM straight to VGPRs and every 16-lane token group keeps
//                  its own online-softmax state (stream_valu).  Several q heads per item (GQA): the
//                  matrix pipe does K.Q^T and V^T.P^T per 32-token tile (stream_mfma).  A GQA group
//                  walks the UNION of its heads' page lists once; a per-page head mask gates each head.
//   * int8 KV    : dequant is fused into the load path: q' = q * scale_k, bytes are converted with
//                  v_cvt_f32_ubyte after flipping the sign bit (u = x + 128), the -128 bias is
//                  folded into the accumulators, out = (P.V / l + offset_v) * scale_v
//                  (sparse_paged_fusion_attention_split_Bbn2s2_Us2.h:2647,2651,4160-4174).
#include "spfa_params.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

namespace spfa {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// readfirstlane makes the wave index provably uniform for the compiler: everything derived from it
// (page cursors, buffer descriptors, loop bounds) then lives in SGPRs and buffer loads need no waterfall loop.
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ unsigned long long lanes_below(int lane) { return (1ull << lane) - 1ull; }

// LDS traffic of ONE wave is in order; this only stops the compiler from caching or moving it.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Data that one workgroup hands to another inside a launch is written and read with agent-scope accesses, which go
// past the non-coherent cache levels, and the writer waits for its stores before it signals with an atomic.  No
// release/acquire fence is used: on this part a fence writes back / invalidates a whole L2, and hundreds of them per
// XCD while the other workgroups still stream K/V cost more than the sharing saves.
template <typename T>
__device__ __forceinline__ void st_agent(T* ptr, T v) { __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ T ld_agent(const T* ptr) {
    return __hip_atomic_load(const_cast<T*>(ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

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
    if constexpr (LPT >= 64) x += __shfl_xor(x, 32);
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
// WAVES: waves of the workgroup that share the centroid tiles (1: the calling wave alone).
// KS = D / 32 MFMA k-steps.  All 16 loads of a group of tiles are issued before the first MFMA (no branch
// sits between them: dtype and KS are template parameters, out-of-range centroids are clamped and masked at
// the store), so one round trip to L2 brings 16 KiB per wave instead of 1 KiB.
// Row r of the tile is q row q_row0 + (r / rpb) * b_stride + r % rpb: rpb consecutive heads of one sequence, then the
// same heads of the next sequence b_stride rows further (rpb = R: all rows consecutive).
// The WAVES waves of a workgroup share the tiles; with nparts > 1 the workgroup is one of nparts that share them (it
// takes every nparts-th group of tiles) and the keys go to global memory with agent-scope stores.
//
// The operand layout of the matrix core wants, per load instruction, 16 bytes of 16 DIFFERENT centroid rows in the
// 16 lanes of a quarter wave: 64 cache lines per instruction, which the CU's address unit works through one line at
// a time (measured: the pass of one workgroup over 512 KiB of centroids took 34 us with two workgroups per CU,
// almost all of it address processing).  So the tiles are read the way they lie in memory, 1 KiB contiguous per
// instruction, parked in a wave-private LDS tile (rows padded by 16 B: the transposed read is conflict free) and
// read back in the operand layout.  `stage` is that tile: score_stage_bytes(D) bytes per wave.
#ifndef SPFA_SCORE_LDS
#define SPFA_SCORE_LDS 1
#endif
constexpr int score_stage_bytes(int D) { return SPFA_SCORE_LDS ? 16 * (D * 2 + 16) : 0; }
__device__ __forceinline__ void wave_order() {   // program order of this wave's LDS accesses (no instruction)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int WAVES, bool BF16, int KS>
__device__ __forceinline__ void score_rows_impl(const void* q, const void* cent, long long q_row0, int n2, int R,
                                                int C, int Cp, uint32_t* keys, char* stage, int wave, int rpb,
                                                int b_stride, int part, int nparts) {
    constexpr int D = KS * 32;
    constexpr int TG = 16 / KS;   // tiles per group: TG * KS = 16 loads in flight
    const int lane = lane_id();
    const int col = lane & 15;   // A: q row of the tile, B: centroid of the tile
    const int kq = lane >> 4;    // which 8-wide slice of the 32-deep k step
#if SPFA_SCORE_LDS
    constexpr int RB = D * 2, RSB = RB + 16;   // bytes of a centroid row, in memory and in the LDS tile
    constexpr int LPR = RB / 16;               // lanes that read one row
    constexpr int RPI = 64 / LPR;              // rows per load instruction (KS instructions per 16-row tile)
    const int lrow = lane / LPR, lpiece = lane % LPR;
    const char* chead = (const char*)cent + (long long)n2 * C * D * 2 + lpiece * 16;
    char* st_w = stage + lrow * RSB + lpiece * 16;
    const char* st_r = stage + col * RSB + kq * 16;
#endif
    uint4 qa[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        qa[kk] = make_uint4(0, 0, 0, 0);
        if (col < R)
            qa[kk] = *(const uint4*)((const char*)q +
                                     ((q_row0 + (long long)(col / rpb) * b_stride + col % rpb) * D + kk * 32 + kq * 8) * 2);
    }
#if !SPFA_SCORE_LDS
    const char* cbase = (const char*)cent + (long long)n2 * C * D * 2 + kq * 16;
#endif
    const int ntiles = Cp >> 4;   // Cp % 128 == 0: a multiple of every TG but the 16 of D = 32 (guarded at the store)
    for (int t0 = (part * WAVES + wave) * TG; t0 < ntiles; t0 += nparts * WAVES * TG) {
        uint4 bv[TG][KS];
#if SPFA_SCORE_LDS
#pragma unroll
        for (int j = 0; j < TG; ++j) {
#pragma unroll
            for (int m = 0; m < KS; ++m) {
                const int c = min((t0 + j) * 16 + m * RPI + lrow, C - 1);
                bv[j][m] = *(const uint4*)(chead + (long long)c * RB);
            }
        }
#else
#pragma unroll
        for (int j = 0; j < TG; ++j) {
            const int c = min((t0 + j) * 16 + col, C - 1);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) bv[j][kk] = *(const uint4*)(cbase + ((long long)c * D + kk * 32) * 2);
        }
#endif
#pragma unroll
        for (int j = 0; j < TG; ++j) {
#if SPFA_SCORE_LDS
            // memory order -> operand order through the wave's LDS tile (LDS accesses of a wave execute in order)
#pragma unroll
            for (int m = 0; m < KS; ++m) *(uint4*)(st_w + m * RPI * RSB) = bv[j][m];
            wave_order();
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) bv[j][kk] = *(const uint4*)(st_r + kk * 64);
            wave_order();
#endif
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                if constexpr (BF16)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, qa[kk]),
                                                                  __builtin_bit_cast(bf16x8, bv[j][kk]), acc, 0, 0, 0);
                else
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, qa[kk]),
                                                                 __builtin_bit_cast(half8, bv[j][kk]), acc, 0, 0, 0);
            }
            // D fragment: row = (lane >> 4) * 4 + r (q head), col = lane & 15 (centroid)
            const int c = (t0 + j) * 16 + col;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qr = kq * 4 + r;
                if (qr < R && c < Cp) {   // a group of TG tiles may reach past the row (D = 32: 16 tiles, Cp % 128 == 0)
                    const uint32_t key = (c < C) ? score_key(acc[r]) : 0u;
                    if (nparts > 1) st_agent(keys + qr * Cp + c, key);
                    else keys[qr * Cp + c] = key;
                }
            }
        }
    }
}

// widx: this wave's index among the WAVES waves that share the tiles.
template <int WAVES>
__device__ __forceinline__ void score_rows(const void* q, const void* cent, int dtype, long long q_row0,
                                           int n2, int R, int D, int C, int Cp, uint32_t* keys, char* stage, int widx,
                                           int rpb = 16, int b_stride = 0, int part = 0, int nparts = 1) {
#define SPFA_SCORE(KS)                                                                              \
    case KS:                                                                                        \
        if (dtype == kBF16)                                                                           \
            score_rows_impl<WAVES, true, KS>(q, cent, q_row0, n2, R, C, Cp, keys, stage, widx, rpb, b_stride, part, nparts);  \
        else                                                                                          \
            score_rows_impl<WAVES, false, KS>(q, cent, q_row0, n2, R, C, Cp, keys, stage, widx, rpb, b_stride, part, nparts); \
        break;
    switch (D >> 5) {
        SPFA_SCORE(1)
        SPFA_SCORE(2)
        SPFA_SCORE(4)
        SPFA_SCORE(8)
        SPFA_SCORE(16)
    }
#undef SPFA_SCORE
}


// The same with the head dim known at compile time (the attention kernels: one D per instantiation).  QDT: kF16 / kBF16
// when the query dtype follows from the KV dtype, -1 when it has to be looked up (int8 KV).
template <int WAVES, int D, int QDT>
__device__ __forceinline__ void score_rows_fixed(const void* q, const void* cent, int dtype, long long q_row0, int n2,
                                                 int R, int C, int Cp, uint32_t* keys, char* stage, int widx,
                                                 int rpb = 16, int b_stride = 0, int part = 0, int nparts = 1) {
    constexpr int KS = D / 32;
    if (QDT == kBF16 || (QDT < 0 && dtype == kBF16))
        score_rows_impl<WAVES, true, KS>(q, cent, q_row0, n2, R, C, Cp, keys, stage, widx, rpb, b_stride, part, nparts);
    else
        score_rows_impl<WAVES, false, KS>(q, cent, q_row0, n2, R, C, Cp, keys, stage, widx, rpb, b_stride, part, nparts);
}

template <int CTRL>
__device__ __forceinline__ int dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true); }
// Sum of v over the 64 lanes (result uniform).
__device__ __forceinline__ int wave_sum(int v) {
    v += dpp_i<0x128>(v);   // row_ror 8
    v += dpp_i<0x124>(v);   // row_ror 4
    v += dpp_i<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_i<0xB1>(v);    // quad_perm [1,0,3,2]
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

// Maximum of v over the 64 lanes (result uniform).
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, (uint32_t)dpp_i<0x128>((int)v));
    v = max(v, (uint32_t)dpp_i<0x124>((int)v));
    v = max(v, (uint32_t)dpp_i<0x4E>((int)v));
    v = max(v, (uint32_t)dpp_i<0xB1>((int)v));
    return max(max((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
               max((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
}

// k-th largest key by a 32-step bitwise bisection: T grows bit by bit while at least k keys are >= T.
// Every lane counts its own keys (compare + add-with-carry, no scalar round trip) and the 64 counts are
// summed once per step.  KeyAt(j) returns this lane's j-th key, j in [0, nj).
template <typename KeyAt>
__device__ __forceinline__ uint32_t bisect_kth(KeyAt key_at, int nj, int k) {
    uint32_t T = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = T | (1u << bit);
        int c = 0;
        for (int j = 0; j < nj; ++j) c += (key_at(j) >= cand) ? 1 : 0;
        if (wave_sum(c) >= k) T = cand;
    }
    return T;
}

// Walk the top-k of one row (Cp % 64 == 0, executed by one full wave): emit(j, key, take, sel) is called for
// every 64-key slice j with this lane's key, whether it belongs to the top-k and the slice's ballot.
// Keys equal to the k-th largest are taken in ascending cluster-id order.  The row in LDS is used as scratch
// and does not survive the call.
template <typename Emit>
__device__ __forceinline__ void topk_walk(uint32_t* krow, int Cp, int k, int lane, Emit emit) {
    const int nj = Cp >> 6;
    auto finish = [&](auto key_at) {
        const uint32_t T = bisect_kth(key_at, nj, k);
        int c = 0;
        for (int j = 0; j < nj; ++j) c += (key_at(j) > T) ? 1 : 0;
        int need = k - wave_sum(c);   // ties at the threshold, lowest cluster ids first
        for (int j = 0; j < nj; ++j) {
            const uint32_t key = key_at(j);
            const unsigned long long eq = __ballot(key == T);
            const bool take = (key > T) || (key == T && (int)__popcll(eq & lanes_below(lane)) < need);
            const unsigned long long sel = __ballot(take);
            need -= min(need, (int)__popcll(eq));
            emit(j, key, take, sel);
        }
    };
    if (nj <= 32) {   // C <= 2048: the row lives in registers
        uint32_t kr[32];
        uint32_t lmax = 0;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            kr[j] = (j < nj) ? krow[j * 64 + lane] : 0u;
            lmax = max(lmax, kr[j]);
        }
        auto count_ge = [&](uint32_t x) {
            int c = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) c += (kr[j] >= x) ? 1 : 0;   // padding keys are 0
            return wave_sum(c);
        };
        // Integer bisection of the k-th largest key T between two bounds with known counts:
        //   count(keys >= lo) = cnt_lo >= k   and   count(keys > hi) = cnt_hi < k.
        // A full step costs a compare per key.  Two things keep the number of full steps small: the k-th largest of
        // the 64 per-lane maxima (a subset of the keys, so never above T) starts lo close to T, and as soon as
        // [lo, hi] holds at most 64 keys they are compacted to one per lane and the rest costs a ballot per step.
        uint32_t hi = wave_max_u32(lmax), lo = 0;
        int cnt_lo = nj * 64, cnt_hi = 0;
        if (k <= 64 && nj > 1) {
            uint32_t a = 0, b = hi;
            for (int s = 0; s < 20 && a < b; ++s) {
                const uint32_t mid = a + ((b - a) >> 1) + 1;
                if ((int)__popcll(__ballot(lmax >= mid)) >= k) a = mid; else b = mid - 1;
            }
            if (a) { lo = a; cnt_lo = count_ge(lo); }
        }
        while (lo < hi && cnt_lo - cnt_hi > 64) {
            const uint32_t mid = lo + ((hi - lo) >> 1) + 1;
            const int c = count_ge(mid);
            if (c >= k) { lo = mid; cnt_lo = c; } else { hi = mid - 1; cnt_hi = c; }
        }
        if (lo < hi) {
            wave_lds_sync();   // every lane holds its keys in registers: the row in LDS becomes the candidate list
            int base = 0;
            const uint32_t span = hi - lo;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const bool in = (kr[j] - lo) <= span && j < nj;
                const unsigned long long bal = __ballot(in);
                if (bal) {
                    if (in) krow[base + (int)__popcll(bal & lanes_below(lane))] = kr[j];
                    base += (int)__popcll(bal);
                }
            }
            wave_lds_sync();
            const bool valid = lane < base;
            const uint32_t mine = valid ? krow[lane] : 0u;
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1) + 1;
                if (cnt_hi + (int)__popcll(__ballot(valid && mine >= mid)) >= k) lo = mid; else hi = mid - 1;
            }
        }
        const uint32_t T = lo;
        int c = 0;
#pragma unroll
        for (int j = 0; j < 32; ++j) c += (kr[j] > T) ? 1 : 0;
        int need = k - wave_sum(c);
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            if (j < nj) {
                const uint32_t key = kr[j];
                const unsigned long long eq = __ballot(key == T);
                const bool take = (key > T) || (key == T && (int)__popcll(eq & lanes_below(lane)) < need);
                const unsigned long long sel = __ballot(take);
                need -= min(need, (int)__popcll(eq));
                emit(j, key, take, sel);
            }
        }
    } else {
        finish([&](int j) { return krow[j * 64 + lane]; });
    }
}

// Top-k of one row as a C-bit bitmap (uint32 words, Cp/32 of them); one full wave.
__device__ __forceinline__ void topk_bitmap(uint32_t* krow, int Cp, int k, uint32_t* bits, int lane) {
    topk_walk(krow, Cp, k, lane, [&](int j, uint32_t, bool, unsigned long long sel) {
        if (lane == 0) {
            bits[2 * j] = (uint32_t)sel;
            bits[2 * j + 1] = (uint32_t)(sel >> 32);
        }
    });
}

// Top-k of one row, sorted descending (ties: ascending id) -> out[0..k); one full wave.
// sortbuf: LDS, npow2 (>= k, power of two) uint64 entries owned by this wave.
__device__ __forceinline__ void topk_sorted(uint32_t* krow, int Cp, int k, int npow2,
                                            unsigned long long* sortbuf, int32_t* out, int lane) {
    int base = 0;
    topk_walk(krow, Cp, k, lane, [&](int j, uint32_t key, bool take, unsigned long long sel) {
        if (take) sortbuf[base + __popcll(sel & lanes_below(lane))] =
            ((unsigned long long)key << 32) | (uint32_t)(~(uint32_t)(j * 64 + lane));
        base += __popcll(sel);
    });
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
    for (int base0 = 0; base0 < page_len; base0 += 256) {
        // both dependent loads (block_table, then block_ids) of four 64-page slices are issued together:
        // two round trips per 256 pages instead of eight
        int raw[4], cid[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base0 + u * 64 + lane;
            raw[u] = (i < page_len) ? bt_row[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base0 + u * 64 + lane;
            const int pg = min(max(raw[u], 0), kv_page_len - 1);   // tests/utils.py:62 clamps
            cid[u] = (i < page_len) ? ids_row[pg] : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int base = base0 + u * 64;
            if (base < page_len) {
                const int i = base + lane;
                const bool in = i < page_len;
                bool hit = in && cid[u] >= 0 && cid[u] < C && ((cbits[cid[u] >> 5] >> (cid[u] & 31)) & 1u);
                if (sink_recent) hit = hit || (in && (raw[u] == bt0 || raw[u] == btm1 || raw[u] == btm2));
                const unsigned long long hb = __ballot(hit);
                const int slot = count + __popcll(hb & lanes_below(lane));
                if (hit && slot < P && pos_row) pos_row[slot] = i;
                if (pbits) {
                    const unsigned long long ab = __ballot(hit && slot < limit);
                    if (lane == 0) pbits[base >> 6] = ab;
                }
                count += __popcll(hb);
            }
        }
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
// One workgroup per (kv head, group of BG sequences): the BG * G q rows that share the head's centroids are
// scored in ONE pass over them (up to 16 MFMA rows), so a centroid slice is read once per group instead of
// once per q row (reference: cent_select.cpp:228-231) or once per sequence.  Groups larger than the key area
// (G > rows_per_pass) fall back to BG = 1 and several passes.  Then one wave per q row.
template <int MODE>   // 0: compute_cent, 1: cent_select
__global__ __launch_bounds__(kThreads) void cent_kernel(const CentParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id(), wave = wave_id();
    const int n2 = blockIdx.x % p.Hkv;   // blocks i and i+8 share an XCD: one kv head's centroids
    const int b0 = (blockIdx.x / p.Hkv) * p.seq_per_group;   // stay in that XCD's L2
    const int RG = p.rows_per_pass;
    uint32_t* keys = (uint32_t*)smem;                                  // [RG][Cp]
    char* aux = smem + (size_t)RG * p.Cp * 4;
    const int nb = min(p.seq_per_group, p.B - b0);                     // sequences of this workgroup
    const int rpb = min(p.G, RG);                                      // rows per sequence in one pass
    for (int g0 = 0; g0 < p.G; g0 += RG) {
        const int Rg = min(RG, p.G - g0);                              // heads of this pass (per sequence)
        const int R = nb * Rg;
        const long long row0 = (long long)b0 * p.Hq + (long long)n2 * p.G + g0;
        score_rows<4>(p.q, p.cent, p.qk_dtype, row0, n2, R, p.D, p.C, p.Cp, keys,
                      smem + p.lds_bytes - (size_t)(4 - wave) * score_stage_bytes(p.D), wave, nb > 1 ? rpb : R, p.Hq);
        __syncthreads();
        for (int r = wave; r < R; r += 4) {
            const int b = b0 + (nb > 1 ? r / rpb : 0);
            const long long row = row0 + (long long)(nb > 1 ? r / rpb : 0) * p.Hq + (nb > 1 ? r % rpb : r);
            if (MODE == 0) {
                int npow2 = 64;
                while (npow2 < p.k) npow2 <<= 1;
                unsigned long long* sortbuf = (unsigned long long*)aux + (size_t)wave * npow2;
                topk_sorted(keys + r * p.Cp, p.Cp, p.k, npow2, sortbuf, p.indices + row * p.k, lane);
            } else {
                int page_len = (p.seq_len[b] + p.block_size - 1) / p.block_size;   // cent_select.cpp:217
                page_len = max(0, min(page_len, p.max_page));
                uint32_t* cbits = (uint32_t*)aux + (size_t)wave * (p.Cp >> 5);
                topk_bitmap(keys + r * p.Cp, p.Cp, p.k, cbits, lane);
                wave_lds_sync();
                const int len = select_pages(cbits, p.C, p.block_table + (long long)b * p.max_page,
                                             p.block_ids + (long long)n2 * p.kv_page_len, p.kv_page_len,
                                             page_len, p.P, p.sink_recent, p.cap_pages,
                                             p.page_position + row * p.P, nullptr, 0, lane);
                if (lane < kLenPad) {
                    p.page_position_length[row * kLenPad + lane] = len;   // cent_select.cpp:418
                    if (len > 0)   // cent_select.cpp:124-152: max_h(len) * page size
                        atomicMax(p.max_page_position_length + (long long)b * kLenPad + lane,
                                  (unsigned long long)len * (unsigned long long)p.block_size);
                }
            }
        }
        __syncthreads();
    }
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
// attention: persistent, wave-specialised kernel
//
//   work item  = (batch, kv head, pass of up to rows_per_pass q heads)   [per-head lists: (batch, q head)]
//   workgroup  = waves 0-3 stream K/V pages of the CURRENT item, the selector wave(s) behind them (one, or a
//                team of two when an item carries several q heads) prepare the NEXT item meanwhile: Q.Cent^T,
//                top-k, page selection, the union page list and the pre-scaled q rows go into the other half
//                of a double-buffered LDS area.
//   One __syncthreads() per item hands the buffers over.  Selection work (latency bound, no HBM
//   traffic) is therefore hidden behind the HBM-bound page stream instead of preceding it.
//   Items are pulled from 8 counters (one per XCD class, item % 8) so that the kv heads an XCD works on
//   keep their centroids in that XCD's L2; a workgroup whose class is exhausted steals from the others,
//   so every item is processed whatever the placement.
// ------------------------------------------------------------------------------------------------
struct PageEntry {            // 8 bytes in LDS
    int phys;                 // physical block
    unsigned short valid;     // attended tokens of the page (<= 512)
    unsigned short mask;      // bit g: head g of the pass attends this page (<= 16 heads per pass)
};
struct ItemHdr {              // 40 bytes in LDS (a 48-byte slot)
    int item;                 // < 0: no more work
    int nU;                   // pages to stream: list[lo .. lo + nU)
    int b, n2;
    int row0;                 // first output row (b * Hq + head) of the pass
    int R;                    // q heads in the pass
    int lo;                   // first list entry of this workgroup's share (page-list split, else 0)
    int next;                 // the item this workgroup handles after this one (claimed one item ahead), < 0: none
    int pc, nch;              // > 0: the workgroup's last item, shared in nch chunks (pc: tail_geometry; see TailBoard)
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

// Build-time tuning knobs (defaults are the shipped configuration).
#ifndef SPFA_MINWAVES
#define SPFA_MINWAVES 4      // launch-bounds waves per SIMD of the one-q-head kernels
#endif
#ifndef SPFA_LG1
#define SPFA_LG1 2           // wave loads of K (and of V) per step, one-q-head kernels: short bursts keep the
                             // CU's memory queue shallow, so the selector wave's dependent loads return quickly
#endif
#ifndef SPFA_LGN
#define SPFA_LGN 4           // the same for the kernels that hold up to 4 q heads (f16/bf16) per pass
#endif
#ifndef SPFA_MINWAVES_N
#define SPFA_MINWAVES_N 2    // launch-bounds waves per SIMD of the 2..4-q-head kernels
#endif
#ifndef SPFA_TAIL_MIN_PAGES
#define SPFA_TAIL_MIN_PAGES 4 // MFMA streamer: fewest pages per chunk of a shared last item
#endif
#ifndef SPFA_MFMA_PREFETCH
#define SPFA_MFMA_PREFETCH 0 // MFMA streamer: 1 = issue the next tile's loads before computing the current one.  Costs
                             // 64 VGPRs, which pushes the selector's top-k keys into scratch: measured slower.
#endif
#ifndef SPFA_MFMA_PREFETCH_I8
#define SPFA_MFMA_PREFETCH_I8 0
#endif
#ifndef SPFA_K_AHEAD
#define SPFA_K_AHEAD 0       // VALU streamer: issue the next step's K loads as soon as this step's K registers are unpacked
#endif
#ifndef SPFA_K_AHEAD_I8
#define SPFA_K_AHEAD_I8 1     // int8 MHA S=64k: 0.799 -> 0.795 ms; f16: no gain
#endif
#ifndef SPFA_PRIO
#define SPFA_PRIO 0          // wave priority of the streamer waves in the item loop (selector stays at 0)
#endif
#ifndef SPFA_KV_AUX
#define SPFA_KV_AUX 2       // cache policy bits of the K/V buffer loads (2 = nt: read once, do not displace the
                             // centroids and tables from L2)
#endif
#ifndef SPFA_WGS_PER_CU
#define SPFA_WGS_PER_CU 2     // resident workgroups per CU of the persistent grid (upper bound)
#endif
// Diagnostic builds (-DSPFA_STAMPS) record s_memtime stamps of the item pipeline into AttnParams::debug;
// the shipped kernel contains none of this.
#ifdef SPFA_STAMPS
#define SPFA_STAMP(slotidx)                                                                         \
    do {                                                                                            \
        if (p.debug && lane == 0)                                                                   \
            p.debug[((size_t)blockIdx.x * 16 + stamp_item) * 16 + (slotidx)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define SPFA_STAMP(slotidx) do {} while (0)
#endif
#ifndef SPFA_MFMA_STREAMERS
#define SPFA_MFMA_STREAMERS 4 // streamer waves of the MFMA kernels (one workgroup per CU fits: 241 VGPRs, LDS)
#endif
// Streamer waves per workgroup.  The one-q-head kernels run two workgroups per CU (8 streamers per CU).
#ifndef SPFA_MFMA_STREAMERS_I8
#define SPFA_MFMA_STREAMERS_I8 6 // int8 tiles are half the bytes: more waves to keep as many bytes in flight (measured
                                 // on GQA 32/8, S=64k: 0.292 -> 0.275 ms; register prefetch instead: slower)
#endif
constexpr int stream_waves(bool mfma, bool i8) { return mfma ? (i8 ? SPFA_MFMA_STREAMERS_I8 : SPFA_MFMA_STREAMERS) : 4; }
// Selector waves per workgroup: one where a single q head is selected per item; two where a pass carries a
// whole GQA group (several rows to select for one page stream), so that selection stays hidden behind the stream.
#ifndef SPFA_MFMA_SELECTORS
#define SPFA_MFMA_SELECTORS 2
#endif
constexpr int sel_waves(bool mfma) { return mfma ? SPFA_MFMA_SELECTORS : 1; }
constexpr int attn_threads(bool mfma, bool i8) { return 64 * (stream_waves(mfma, i8) + sel_waves(mfma)); }

// Pull the next work item.  Class c holds items c, c+8, ...; start with this XCD's class.
// The first item of every workgroup is static (item = blockIdx.x, which is the (blockIdx.x / 8)-th item of
// class blockIdx.x % 8), so the counters only hand out the items behind those.
// Classes are tried in the order xcd, xcd + 1, ... (from the t_first-th on).  One look at all eight counters tells
// which classes still have items, so an exhausted queue costs one load instead of eight dependent atomics.
__device__ __forceinline__ int fetch_item(unsigned* counters, int xcd, int total, int grid, int lane, int t_first = 0) {
    const int c = (xcd + lane) & 7;
    const int n_c = (total - c + 7) >> 3;      // items of class c
    const int first = (grid - c + 7) >> 3;     // of which statically assigned
    bool open = false;
    if (lane >= t_first && lane < 8 && n_c > first)
        open = __hip_atomic_load(&counters[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + (unsigned)first < (unsigned)n_c;
    unsigned long long cand = __ballot(open);
    int item = -1;
    while (cand && item < 0) {
        const int t = __ffsll((long long)cand) - 1;
        cand &= cand - 1;
        const int ct = (xcd + t) & 7;
        const int n_t = (total - ct + 7) >> 3, first_t = (grid - ct + 7) >> 3;
        unsigned i = 0;
        if (lane == 0) i = atomicAdd(&counters[ct], 1u) + (unsigned)first_t;
        i = (unsigned)__builtin_amdgcn_readfirstlane((int)i);
        if (i < (unsigned)n_t) item = ct + 8 * (int)i;
    }
    return item;
}
// The same in two halves, so that the round trip of the first atomic (own class) overlaps the work between them:
// fetch_begin issues it and returns the raw counter value in lane 0, fetch_end interprets it and only falls back to
// the other classes (blocking) when the own class is exhausted.
__device__ __forceinline__ unsigned fetch_begin(unsigned* counters, int xcd, int total, int grid, int lane) {
    const int n_c = (total - xcd + 7) >> 3, first = (grid - xcd + 7) >> 3;
    unsigned raw = 0xffffffffu;
    if (lane == 0 && n_c > first) raw = atomicAdd(&counters[xcd], 1u);
    return raw;
}
__device__ __forceinline__ int fetch_end(unsigned raw, unsigned* counters, int xcd, int total, int grid, int lane) {
    raw = (unsigned)__builtin_amdgcn_readfirstlane((int)raw);
    const int n_c = (total - xcd + 7) >> 3, first = (grid - xcd + 7) >> 3;
    if (raw != 0xffffffffu && raw + (unsigned)first < (unsigned)n_c) return xcd + 8 * (int)(raw + (unsigned)first);
    return fetch_item(counters, xcd, total, grid, lane, 1);
}

// ------------------------------------------------------------------------------------------------
// Sharing the last item of a workgroup.
//
// The items of a launch are equal-sized only on paper: XCDs run a few percent apart and real page lists differ in
// length, and because the next item is claimed one item ahead nothing is left to steal when the fast workgroups run
// dry.  So the LAST item of every workgroup is not streamed by static striding but in chunks of a few pages claimed
// from a word in global memory: the owner's waves claim chunks from the front, workgroups that have finished their
// own work claim from the back (they read the page list, which the owner published next to the word).  Every party
// leaves an un-normalised state in the workspace and adds its chunk count to an arrival counter; the party that
// completes the count merges the states and stores the output rows.  Nobody ever waits for anybody.
// ------------------------------------------------------------------------------------------------
struct TailBoard {           // 16 bytes per workgroup, zeroed before every launch
    unsigned state;          // front (chunks claimed by the owner) | back << 16 (chunks below `back` are unclaimed)
    unsigned pad;
    unsigned long long done; // low word: chunks accounted for, + 1 for the owner's arrival; high word: bit c set when
                             // a helper party left its state in slot 1 + c (c = the lowest chunk it took)
};
struct TailHdr {             // 32 bytes at the start of a board's data area, written before the board is offered
    int nU, pc, nch;         // list length, chunk geometry (tail_geometry), chunks
    int n2, row0, R, b, pad;
};
constexpr int kHdrSlot = 48;                        // bytes per ItemHdr slot in LDS
constexpr int kTailChunksMax = 32;                  // chunks per shared item (AttnParams::tail_chunks <= this)

// Chunks of a shared item: the first nbig chunks hold pc pages each, the rest ps <= pc pages.  The owner claims from
// the front, where a claim (which drains the wave's loads in flight) buys a long run of pages; the back, where the
// helpers arrive and where the owner's waves finish, is cut finely: about the last third of the list goes into half
// of the chunks, so what one party still holds when everything is claimed is a small fraction of the item.
// Returns pc | ps << 12 | nbig << 24 and the chunk count (<= max_chunks <= 32); list lengths stay below 4096.
__device__ __forceinline__ int tail_geometry(int nU, int max_chunks, int min_pc, int& nch) {
    if (nU <= max_chunks) { nch = nU; return 1 | (1 << 12); }
    const int ns = max_chunks / 2;
    const int ps = max(1, (nU / 3 + ns - 1) / ns);
    const int big_pages = nU - ns * ps;
    const int pc = max(min_pc, (big_pages + (max_chunks - ns) - 1) / (max_chunks - ns));
    const int nbig = min((big_pages + pc - 1) / pc, nU / pc);
    nch = nbig + (nU - nbig * pc + ps - 1) / ps;
    return pc | (ps << 12) | (nbig << 24);
}

// Page cursor of one streamer wave: static striding (wave, wave + 4, ...) or chunk claims from a board.
struct PageCursor {
    unsigned* state;   // board word (modes 1, 2)
    int mode;          // 0: static, 1: owner (front), 2: helper (back)
    int pc, ps, nbig, nch, nU;     // chunk geometry, see tail_geometry()
    int n, n_end, chunks, first;   // first: lowest chunk claimed
    int stride;                    // streamer waves of the workgroup (static striding)
    __device__ __forceinline__ void init(int mode_, unsigned* state_, int geom, int nch_, int nU_, int wave, int stride_) {
        mode = mode_; state = state_; nch = nch_; nU = nU_; stride = stride_;
        pc = geom & 0xfff; ps = (geom >> 12) & 0xfff; nbig = geom >> 24;
        n = mode_ == 0 ? wave - stride_ : 0; n_end = 0; chunks = 0; first = 1 << 20;
    }
    __device__ __forceinline__ int claim(int lane) {
        int c = -1;
        if (lane == 0) {
            if (mode == 1) {
                const unsigned old = atomicAdd(state, 1u);
                const int f = (int)(old & 0xffffu), bk = (int)(old >> 16);
                if (f < bk && f < nch) c = f;
            } else {
                for (int tries = 0; tries < 16; ++tries) {
                    const unsigned old = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int f = (int)(old & 0xffffu), bk = (int)(old >> 16);
                    if (bk <= f || bk > nch) break;
                    if (atomicCAS(state, old, old - 0x10000u) == old) { c = bk - 1; break; }
                }
            }
        }
        return __builtin_amdgcn_readfirstlane(c);
    }
    // advance to the next page of this wave; false when there is none
    __device__ __forceinline__ bool next(int lane) {
        if (mode == 0) { n += stride; return n < nU; }
        if (++n < n_end) return true;
        const int c = claim(lane);
        if (c < 0) return false;
        ++chunks;
        first = min(first, c);
        n = c < nbig ? c * pc : nbig * pc + (c - nbig) * ps;
        n_end = min(nU, n + (c < nbig ? pc : ps));
        return n < n_end;
    }
};

// ------------------------------------------------------------------------------------------------
// Streamer, VALU form (one q head per pass, and int8 KV): K/V pages of one item -> this wave's partial
// softmax state in LDS.  list / qst / comb are the item's page list, pre-scaled q rows [NQ][D] and the
// [4 waves][NQ][D + 2] state area.
// ------------------------------------------------------------------------------------------------
template <typename KVT, int NQ, int LPT>
__device__ __forceinline__ void stream_valu(const AttnParams& p, const PageEntry* list, const float* qst, float* comb,
                                            PageCursor& cur, int n2, long long row0, int n_base, int lane, int wave) {
    constexpr int EPL = KvTraits<KVT>::EPL;     // elements per 16-byte lane load
    constexpr int TPL = 64 / LPT;               // tokens per wave load
    constexpr int LG = (NQ * EPL <= 16) ? SPFA_LG1 : ((NQ * EPL <= 32) ? SPFA_LGN : 2);   // wave loads of K (and V) per step
    constexpr int STEP = TPL * LG;
    constexpr bool QUANT = sizeof(KVT) == 1;
    constexpr bool K_AHEAD = QUANT ? SPFA_K_AHEAD_I8 : SPFA_K_AHEAD;
    constexpr float kNegBig = -1.0e30f;
    const int sub = lane % LPT;                 // 16-byte slice of the head row
    const int grp = lane / LPT;                 // token within a wave load
    const int D = p.D;
    const int cstride = D + 2;
    float qf[NQ][EPL];
    float qbias[NQ];   // int8: 128 * sum of this lane's q' (u = x + 128)
#pragma unroll
    for (int g = 0; g < NQ; ++g) {
        qbias[g] = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            qf[g][e] = qst[g * D + sub * EPL + e];
            qbias[g] += 128.f * qf[g][e];
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
    // Addressing: one buffer descriptor per page and head (uniform, SGPRs) + a per-lane byte offset
    // that never changes (token group * token stride + 16-byte slice) + a scalar offset per wave
    // load.  The descriptor's range ends at the last live row of this head in the page, so token slots past the
    // page's live tokens read zeros instead of faulting or picking up whatever the cache holds there (a NaN
    // bit pattern in a dead V row would otherwise poison 0 * v).
    const int lane_off = grp * (int)p.tok_stride + sub * 16;
    const int row_bytes = D * (int)sizeof(KVT);
    // Score bias / mask over the columns of this head's gathered sequence (one head per pass whenever either is
    // given): column = list slot * block_size + token (…Us2.h:2192-2194,2239-2244,3366-3404).  Scores are kept in
    // base-2 exponent units here, so the bias is scaled like them; a masked score becomes -1e12 (…Us2.h:860).
    constexpr float kLog2e = 1.4426950408889634f;
    constexpr float kMasked = -1.0e12f * kLog2e;
    const bool biased = NQ == 1 && (p.pse != nullptr || p.amask != nullptr);
    const int b_of_row = biased ? (int)(row0 / p.Hq) : 0;
    const long long pse_row = ((long long)(p.pse_b == 1 ? 0 : b_of_row) * p.Hq + (row0 - (long long)b_of_row * p.Hq)) * p.pse_s;
    const long long mask_row = (long long)b_of_row * p.amask_s;
    while (cur.next(lane)) {
        const int n = cur.n;
        const int phys = __builtin_amdgcn_readfirstlane(list[n].phys);
        const int valid = __builtin_amdgcn_readfirstlane(list[n].valid);
        const unsigned mask = __builtin_amdgcn_readfirstlane((unsigned)list[n].mask);
        const long long off = (long long)phys * p.blk_stride + (long long)n2 * p.head_stride;
        const int page_range = (valid - 1) * (int)p.tok_stride + row_bytes;
        const __amdgpu_buffer_rsrc_t rk =
            __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.kcache + off), 0, page_range, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv =
            __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.vcache + off), 0, page_range, 0x00020000);
        uint4 kreg[LG], vreg[LG];
        auto load_k = [&](int t0) {
#pragma unroll
            for (int i = 0; i < LG; ++i)
                kreg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                                        rk, lane_off, (t0 + i * TPL) * (int)p.tok_stride, SPFA_KV_AUX));
        };
        if (K_AHEAD) load_k(0);
        for (int t0 = 0; t0 < valid; t0 += STEP) {
            if (!K_AHEAD) load_k(t0);
#pragma unroll
            for (int i = 0; i < LG; ++i)
                vreg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                                        rv, lane_off, (t0 + i * TPL) * (int)p.tok_stride, SPFA_KV_AUX));
            float s[NQ][LG];
            float kf[LG][EPL];
#pragma unroll
            for (int i = 0; i < LG; ++i) unpack16<KVT>(kreg[i], kf[i]);
            // the K registers are free again: the next step's K rows travel while this step's softmax and V rows are
            // worked on (rows past the page end read as zeros and are never used)
            if (K_AHEAD && t0 + STEP < valid) load_k(t0 + STEP);
#pragma unroll
            for (int i = 0; i < LG; ++i) {
                const bool live = (t0 + i * TPL + grp) < valid;
#pragma unroll
                for (int g = 0; g < NQ; ++g) {
                    float acc = QUANT ? -qbias[g] : 0.f;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) acc = fmaf(qf[g][e], kf[i][e], acc);
                    acc = group_allsum<LPT>(acc);
                    if (biased && live) {
                        const long long col = (long long)(n_base + n) * p.block_size + (t0 + i * TPL + grp);
                        if (p.pse && col < p.pse_s) acc += load_qk_elem(p.pse, p.q_dtype, pse_row + col) * kLog2e;
                        if (p.amask && col < p.amask_s && p.amask[mask_row + col]) acc = kMasked;
                    }
                    s[g][i] = live ? acc : -INFINITY;
                }
            }
#pragma unroll
            for (int g = 0; g < NQ; ++g) {
                if (!((mask >> g) & 1u)) {   // wave-uniform: this head does not attend the page
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
                    if ((mask >> g) & 1u) {
#pragma unroll
                        for (int e = 0; e < EPL; ++e) o[g][e] = fmaf(s[g][i], vf[e], o[g][e]);
                    }
                }
            }
        }
    }
    // merge the 64/LPT token groups of the wave; the 4 waves are merged after the hand-over barrier
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
        float* dst = comb + ((size_t)wave * NQ + g) * cstride;
        if (grp == 0) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) dst[sub * EPL + e] = o[g][e];
            if (sub == 0) { dst[D] = m[g]; dst[D + 1] = l[g]; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Streamer, MFMA form (2..16 q heads of a group share every K/V row; f16 / bf16 KV, or int8 KV converted to f16
// on the way into the matrix pipe).
//
// Per 32-token tile of a page, one wave computes with v_mfma_f32_16x16x32:
//   S^T[token][head] = K . Q^T     A = K rows straight from HBM (lane: token l&15, 16-byte slice l>>4 of each
//                                  32-element k step), B = the q heads padded to 16 columns (loaded once per item);
//   online softmax   in the accumulator layout: a lane holds 8 tokens (4 of each 16-token half) of head l&15;
//                    the running maximum is shared by the 4 lanes of a head (two cross-lane exchanges per tile),
//                    the running sum stays per lane until the end;
//   O^T[d][head]    += V^T . P^T   B = P^T taken from the accumulator registers as they are (the k order of a step
//                                  is free as long as both operands agree: k slot (l>>4, j) is token 4(l>>4)+j of the
//                                  first half for j < 4 and of the second half for j >= 4), A = V^T read with
//                                  ds_read_b64_tr_b16 from a row-major image of the tile in LDS (rows padded by
//                                  32 B: the 8 rows a 32-lane half touches fall into 8 different bank groups).
// V goes HBM -> VGPR -> LDS (wave-private area, so no workgroup barrier), K never touches LDS.  P is split into a high and a low f16 / bf16 part (two
// MFMAs, the matrix pipe is nearly idle anyway) so that its rounding stays far below the output tolerance.
// ------------------------------------------------------------------------------------------------
template <typename KVT, int LPT>
__device__ __forceinline__ void stream_mfma(const AttnParams& p, const PageEntry* list, float* comb, char* vst,
                                            PageCursor& cur, int n2, long long row0, int R, int nq, float qscale,
                                            int lane, int wave) {
    constexpr bool BF16 = __is_same(KVT, __bf16);
    constexpr bool I8 = sizeof(KVT) == 1;
    constexpr int ES = (int)sizeof(KVT);
    constexpr int D = LPT * 16 / ES;
    constexpr int KS = D / 32;                  // MFMA k steps of Q.K^T
    constexpr int KL = I8 ? KS / 2 : KS;        // 16-byte K loads per lane and 16-token half
    constexpr int DT = D / 16;                  // 16-row tiles of O^T
    constexpr int TPL = 64 / LPT;               // V rows per wave load
    constexpr int VL = 32 / TPL;                // V wave loads per tile
    constexpr int RS = D * 2 + 32;              // LDS row stride of the (16-bit) V image
    constexpr bool PREFETCH = (I8 ? SPFA_MFMA_PREFETCH_I8 : SPFA_MFMA_PREFETCH) && LPT <= 16;   // two tiles of registers fit
    constexpr float kNegBig = -1.0e30f;
    typedef short short4v __attribute__((ext_vector_type(4)));
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    const int h = lane & 15, sg = lane >> 4;
    const int cstride = D + 2;

    // int8 -> f16, exact: flip the sign bit (u = x + 128), splice each byte under the exponent of 1024.0
    // (0x6400 | u = 1024 + u) and subtract 1152.  One dword (4 values) -> two packed-f16 dwords in 5 VALU ops.
    auto cvt4 = [](uint32_t w, uint32_t& lo, uint32_t& hi) {
        const uint32_t u = w ^ 0x80808080u;
        const half2v bias = {(_Float16)1152.f, (_Float16)1152.f};
        lo = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2v, __builtin_amdgcn_perm(0x64646464u, u, 0x04010400u)) - bias);
        hi = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2v, __builtin_amdgcn_perm(0x64646464u, u, 0x04030402u)) - bias);
    };
    auto cvt16 = [&](const uint4& v, uint4& a, uint4& b) {   // 16 int8 -> two 8 x f16 fragments, element order kept
        cvt4(v.x, a.x, a.y); cvt4(v.y, a.z, a.w);
        cvt4(v.z, b.x, b.y); cvt4(v.w, b.z, b.w);
    };

    // B operand of K.Q^T: q head h (column), this lane's 8 elements of every k step.  f16 / bf16: the q row as it is,
    // element kk * 32 + sg * 8 + j.  int8 KV: q' = q * scale_k per channel (…Us2.h:2647) as a high and a low f16
    // part; a lane's K bytes are the 32 consecutive channels sg * (D / 4) ..., so step kk takes channels
    // (kk / 2) * 64 + sg * 16 + (kk % 2) * 8 + j.
    uint4 qf[KS], qlo[I8 ? KS : 1];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        qf[kk] = make_uint4(0, 0, 0, 0);
        if constexpr (I8) {
            qlo[kk] = make_uint4(0, 0, 0, 0);
            if (h < R) {
                const int d0 = (kk >> 1) * 64 + sg * 16 + (kk & 1) * 8;
                half8 hi8, lo8;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = load_qk_elem(p.q, p.q_dtype, (row0 + h) * D + d0 + j) *
                                    load_qk_elem(p.antiq_scale, p.q_dtype, p.antiq_per_tensor ? 0 : (long long)n2 * D + d0 + j);
                    hi8[j] = (_Float16)x;
                    lo8[j] = (_Float16)(x - (float)hi8[j]);
                }
                qf[kk] = __builtin_bit_cast(uint4, hi8);
                qlo[kk] = __builtin_bit_cast(uint4, lo8);
            }
        } else {
            if (h < R) qf[kk] = *(const uint4*)((const char*)p.q + ((row0 + h) * D + kk * 32 + sg * 8) * 2);
        }
    }
    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = kNegBig, l = 0.f;

    const int k_lane_off = h * (int)p.tok_stride + sg * 16;
    const int v_lane_off = (lane / LPT) * (int)p.tok_stride + (lane % LPT) * 16;
    char* v_wr = vst + (lane / LPT) * RS + (lane % LPT) * (I8 ? 32 : 16);
    // transposed read: lane 4q+pp of a 16-lane group addresses row q, columns 4pp..4pp+3 of the 16-column block
    const char* v_rd = vst + (4 * sg + (h >> 2)) * RS + (h & 3) * 8;

    struct Tile { uint4 k[2][KL]; uint4 v[VL]; };
    auto issue = [&](Tile& T, const __amdgpu_buffer_rsrc_t& rk, const __amdgpu_buffer_rsrc_t& rv, int t0) {
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int kk = 0; kk < KL; ++kk)
                T.k[half][kk] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                    rk, k_lane_off + kk * 64, (t0 + 16 * half) * (int)p.tok_stride, SPFA_KV_AUX));
#pragma unroll
        for (int i = 0; i < VL; ++i)
            T.v[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                rv, v_lane_off, (t0 + i * TPL) * (int)p.tok_stride, SPFA_KV_AUX));
    };
    auto mfma16 = [](const uint4& a, const uint4& b, f32x4 c) {
        if constexpr (BF16)
            return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
        else
            return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    };
    auto compute = [&](const Tile& T, int t0, int valid, bool att) {
        f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
        if constexpr (I8) {
            f32x4 sa2 = {0.f, 0.f, 0.f, 0.f}, sb2 = {0.f, 0.f, 0.f, 0.f};   // the low parts: two more independent chains
#pragma unroll
            for (int kl = 0; kl < KL; ++kl) {
                uint4 a0, a1, b0, b1;
                cvt16(T.k[0][kl], a0, a1);
                cvt16(T.k[1][kl], b0, b1);
                sa = mfma16(a0, qf[2 * kl], sa);      sa2 = mfma16(a0, qlo[2 * kl], sa2);
                sb = mfma16(b0, qf[2 * kl], sb);      sb2 = mfma16(b0, qlo[2 * kl], sb2);
                sa = mfma16(a1, qf[2 * kl + 1], sa);  sa2 = mfma16(a1, qlo[2 * kl + 1], sa2);
                sb = mfma16(b1, qf[2 * kl + 1], sb);  sb2 = mfma16(b1, qlo[2 * kl + 1], sb2);
            }
            sa += sa2;
            sb += sb2;
        } else {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                sa = mfma16(T.k[0][kk], qf[kk], sa);
                sb = mfma16(T.k[1][kk], qf[kk], sb);
            }
        }
        // V image of the tile (16-bit, row major): the previous tile's transposed reads were issued earlier by this
        // same wave and LDS executes a wave's operations in order, so the area can be overwritten right away
#pragma unroll
        for (int i = 0; i < VL; ++i) {
            if constexpr (I8) {
                uint4 a, b;
                cvt16(T.v[i], a, b);
                *(uint4*)(v_wr + i * TPL * RS) = a;
                *(uint4*)(v_wr + i * TPL * RS + 16) = b;
            } else {
                *(uint4*)(v_wr + i * TPL * RS) = T.v[i];
            }
        }
        float s[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ta = t0 + 4 * sg + r;
            s[r] = (att && ta < valid) ? sa[r] * qscale : -INFINITY;
            s[4 + r] = (att && ta + 16 < valid) ? sb[r] * qscale : -INFINITY;
        }
        float cmax = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7])));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 16));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
        const float mn = fmaxf(m, cmax);
        const float alpha = fast_exp2(m - mn);
        m = mn;
        float pr[8], psum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { pr[j] = fast_exp2(s[j] - mn); psum += pr[j]; }
        l = l * alpha + psum;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] *= alpha;
        asm volatile("" ::: "memory");   // the image is complete (in program order) before it is read back
        __builtin_amdgcn_wave_barrier();
        uint4 ph, pl;
        if constexpr (BF16) {
            bf16x8 x, y;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                x[j] = (__bf16)pr[j];
                y[j] = (__bf16)(pr[j] - (float)x[j]);
            }
            ph = __builtin_bit_cast(uint4, x);
            pl = __builtin_bit_cast(uint4, y);
        } else {
            half8 x, y;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                x[j] = (_Float16)pr[j];
                y[j] = (_Float16)(pr[j] - (float)x[j]);
            }
            ph = __builtin_bit_cast(uint4, x);
            pl = __builtin_bit_cast(uint4, y);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const short4v va = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) short4v*)(v_rd + dt * 32));
            const short4v vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) short4v*)(v_rd + 16 * RS + dt * 32));
            const uint4 vv = make_uint4(__builtin_bit_cast(uint2, va).x, __builtin_bit_cast(uint2, va).y,
                                        __builtin_bit_cast(uint2, vb).x, __builtin_bit_cast(uint2, vb).y);
            o[dt] = mfma16(vv, ph, o[dt]);
            o[dt] = mfma16(vv, pl, o[dt]);
        }
        asm volatile("" ::: "memory");
    };

    const int row_bytes = D * ES;
    // pages come from the cursor (static striding or chunk claims), 32-token tiles inside a page
    int t0 = 0, valid = 0;
    bool att = false;
    __amdgpu_buffer_rsrc_t rk, rv;
    auto open_page = [&]() {
        const int n = cur.n;
        const int phys = __builtin_amdgcn_readfirstlane(list[n].phys);
        valid = __builtin_amdgcn_readfirstlane((int)list[n].valid);
        const unsigned mask = __builtin_amdgcn_readfirstlane((unsigned)list[n].mask);
        att = (mask >> h) & 1u;
        const long long off = (long long)phys * p.blk_stride + (long long)n2 * p.head_stride;
        const int range = (valid - 1) * (int)p.tok_stride + row_bytes;   // dead rows of the page read as zeros
        rk = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.kcache + off), 0, range, 0x00020000);
        rv = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.vcache + off), 0, range, 0x00020000);
    };
    if (cur.next(lane)) {
        open_page();
        Tile cur_t;
        issue(cur_t, rk, rv, 0);
        while (true) {
            const int c_t0 = t0, c_valid = valid;
            const bool c_att = att;
            bool more = true;
            t0 += 32;
            if (t0 >= valid) { t0 = 0; more = cur.next(lane); if (more) open_page(); }
            if constexpr (PREFETCH) {
                Tile nxt;
                if (more) issue(nxt, rk, rv, t0);
                compute(cur_t, c_t0, c_valid, c_att);
                if (!more) break;
                cur_t = nxt;
            } else {
                compute(cur_t, c_t0, c_valid, c_att);
                if (!more) break;
                issue(cur_t, rk, rv, t0);
            }
        }
    }
    // this wave's state: the sum over its 4 lanes per head, O^T[d = 16 dt + 4 sg + r][head h]
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (h < R) {
        float* dst = comb + ((size_t)wave * nq + h) * cstride;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            *(f32x4*)(dst + dt * 16 + 4 * sg) = o[dt];
        if (sg == 0) { dst[D] = m; dst[D + 1] = l; }
    }
}

// NQ: q heads per pass of the VALU streamer (register arrays); the MFMA streamer takes p.rows_per_pass (<= 16).
template <typename KVT, int NQ, int LPT, bool MFMA>
__global__ __launch_bounds__(attn_threads(MFMA, sizeof(KVT) == 1), (MFMA ? 2 : (NQ == 1 ? SPFA_MINWAVES : (NQ <= 4 ? SPFA_MINWAVES_N : 1))))
void attn_kernel(const AttnParams p) {
    constexpr int SW = sel_waves(MFMA);           // selector team size
    constexpr int kStreamWaves = stream_waves(MFMA, sizeof(KVT) == 1);
    constexpr int WAVES = kStreamWaves + SW;
    constexpr bool QUANT = sizeof(KVT) == 1;
    constexpr int DC = LPT * 16 / (int)sizeof(KVT);   // head dim of this instantiation (== p.D)
    constexpr int QDT = QUANT ? -1 : (__is_same(KVT, __bf16) ? (int)kBF16 : (int)kF16);
    constexpr float kNegBig = -1.0e30f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id(), wave = wave_id();
    const bool selector = wave >= kStreamWaves;
    const int sidx = wave - kStreamWaves;         // index inside the selector team (leader = 0)
    const int D = p.D;
    const int nq = MFMA ? p.rows_per_pass : NQ;   // row capacity of a pass (LDS areas are sized by it)
    const bool per_head = (p.mode == 0 && p.pos_rows == p.Hq && p.G > 1);   // every q head has its own list
    const int G_eff = per_head ? 1 : p.G;
    const int n_pass = (G_eff + nq - 1) / nq;
    const int heads_per_b = per_head ? p.Hq : p.Hkv;
    // Page-list split (small batches): every item is dealt to NS workgroups, each repeats the (deterministic)
    // selection, streams one NS-th of the list and leaves an un-normalised state in global memory; the last
    // of them to arrive merges the NS states.  Then there is exactly one unit per workgroup.
    const int NS = p.n_split;
    const int total = p.B * heads_per_b * n_pass * NS;

    uint32_t* keys = (uint32_t*)(smem + p.off_keys);                              // [nq][Cp]   (selector)
    uint32_t* cbits = (uint32_t*)(smem + p.off_cbits);                            // [min(nq,5)][Cp/32]
    int* row_lens = (int*)(smem + p.off_pbits) - 16;                              // [16] list lengths of the first item
    unsigned long long* pbits_all = (unsigned long long*)(smem + p.off_pbits);    // [nq][n_pwords] (selector)
    PageEntry* lists = (PageEntry*)(smem + p.off_list);                           // [2][max_entries]
    float* combs = (float*)(smem + p.off_comb);                                   // [2][4][nq][D+2]
    float* qsts = (float*)(smem + p.off_qst);                                     // [2][nq][D]
    ItemHdr* hdrs = (ItemHdr*)(smem + p.off_hdr);                                 // [2]
    int* team = (int*)(smem + p.off_hdr + 2 * kHdrSlot);   // [0] selector team counter, [1] lowest chunk, [2] flag, [3] chunks
    auto hdr_at = [&](int slot) { return (ItemHdr*)((char*)hdrs + slot * kHdrSlot); };
    // LDS tile of wave w for the centroid pass: the MFMA streamers lend their V tile area (only the first item is
    // scored by them, before anything is streamed), everybody else has a tile at off_stage
    auto stage_of = [&](int w) -> char* {
        if (MFMA && w < kStreamWaves) return smem + p.off_vst + (size_t)w * (32 * (D * 2 + 32));
        return smem + p.off_stage + (size_t)(MFMA ? w - kStreamWaves : w) * score_stage_bytes(D);
    };
    TailBoard* boards = (TailBoard*)p.tail_boards;
    TailBoard* my_board = boards + blockIdx.x;
    auto board_data = [&](int v) { return p.tail_data + (size_t)v * p.tail_stride; };
    // data area of a board: TailHdr | page list [max_entries] | pre-scaled q rows [nq][D] | states [1 + tail_chunks][nq][D+2]
    const size_t tail_list_off = sizeof(TailHdr);
    const size_t tail_qst_off = tail_list_off + (size_t)p.max_entries * sizeof(PageEntry);
    const size_t tail_part_off = (tail_qst_off + (MFMA ? 0 : (size_t)nq * D * 4) + 15) & ~(size_t)15;
    int team_epoch = 0;
    // Meeting point of the selector team (not of the workgroup): a counter in LDS.  Every team wave passes through
    // the same sequence of calls, so the spin always ends; LDS operations of a wave execute in order, which makes
    // everything it wrote before the increment visible to the wave that sees the count.
    auto team_sync = [&]() {
        wave_lds_sync();
        if constexpr (SW > 1) {
            team_epoch += SW;
            if (lane == 0) atomicAdd(team, 1);
            while (*(volatile int*)team < team_epoch) __builtin_amdgcn_s_sleep(1);
            wave_lds_sync();
        }
    };
    const int n_pwords = (p.max_page + 63) >> 6;
    const int cstride = D + 2;
    const float qscale = p.scale * 1.4426950408889634f;   // scores in base-2 exponent units
    // Workgroups are dealt round-robin over the 8 XCDs, so blockIdx % 8 labels the workgroups that share an
    // XCD (and its L2).  Used for L2 affinity of the centroids only, never for correctness.
    const int xcd = blockIdx.x & 7;
#ifdef SPFA_STAMPS
    int stamp_item = 0;
#endif

    // ---------------- selector: everything that precedes the page stream of one item ----------------
    struct ItemCtx { int b, n2, head0, R, split; long long row0; };
    auto decode = [&](int unit) {
        // unit -> (batch, head, pass, split); unit % 8 picks the XCD class, so keep the kv head in the low part
        ItemCtx c;
        c.split = unit % NS;
        const int item = unit / NS;
        const int hh = item % heads_per_b;
        const int rest = item / heads_per_b;
        const int pass = rest % n_pass;
        c.b = rest / n_pass;
        c.n2 = per_head ? hh / p.G : hh;
        c.head0 = per_head ? hh : c.n2 * p.G + pass * nq;
        c.R = min(nq, G_eff - pass * nq);
        c.row0 = (long long)c.b * p.Hq + c.head0;
        return c;
    };
    // top-k clusters of row r of an item -> its page list (outputs + LDS bitmap); returns the list length
    auto select_row = [&](const ItemCtx& ctx, int r, uint32_t* cb, int page_len) {
        topk_bitmap(keys + r * p.Cp, p.Cp, p.k, cb, lane);
        wave_lds_sync();
        SPFA_STAMP(3);
        const long long row = ctx.row0 + r;
        const int len = select_pages(cb, p.C, p.block_table + (long long)ctx.b * p.max_page,
                                     p.block_ids + (long long)ctx.n2 * p.kv_page_len, p.kv_page_len, page_len, p.P,
                                     p.sink_recent, p.cap_pages,
                                     (p.page_position && ctx.split == 0) ? p.page_position + row * p.P : nullptr,
                                     pbits_all + (size_t)r * n_pwords, n_pwords, lane);
        if (p.page_position_length && ctx.split == 0 && lane < kLenPad)
            p.page_position_length[row * kLenPad + lane] = len;
        wave_lds_sync();
        return len;
    };
    auto page_len_of = [&](int b) {
        return max(0, min((p.seq_len[b] + p.block_size - 1) / p.block_size, p.max_page));
    };
    // item < 0: fetch the next one; stage: 0 nothing done yet, 1 the keys of this item are already in LDS,
    // 2 the rows' page lists are done as well (bitmaps in pbits_all, lengths in row_lens)
    // raw_next: the value of fetch_begin when the caller has issued it already (first item), else kNoFetch
    constexpr unsigned kNoFetch = 0xfffffffeu;
    int fetch_round = 0;   // claims made by this workgroup so far
    auto build = [&](int slot, int item, int stage, unsigned raw_next) {
        ItemHdr* hdr = hdr_at(slot);
        PageEntry* list = lists + (size_t)slot * p.max_entries;
        float* qst = qsts + (size_t)slot * nq * D;
        SPFA_STAMP(0);
        if (item < 0) {
            if (lane == 0 && sidx == 0) hdr->item = -1;
            return;
        }
        // claim the item after this one now (the leader; the answer is needed at the end): whether this is the
        // workgroup's last item must be known before it is streamed
        const bool dynamic = total > (int)gridDim.x;
        // class whose queue this claim tries first: the own XCD's, or (class_step = 1) one class further per claim
        const int cls = (xcd + p.class_step * (++fetch_round)) & 7;
        if (raw_next == kNoFetch && dynamic && sidx == 0)
            raw_next = fetch_begin(p.counters, cls, total, (int)gridDim.x, lane);
        const ItemCtx ctx = decode(item);
        const int b = ctx.b, n2 = ctx.n2, head0 = ctx.head0, R = ctx.R;
        const long long row0 = ctx.row0;
        const int32_t* bt_row = p.block_table + (long long)b * p.max_page;
        int nU = 0;
        if (p.mode == 1) {
            const int seq = p.seq_len[b];
            const int page_len = page_len_of(b);
            if (stage < 1)
                score_rows_fixed<SW, DC, QDT>(p.q, p.cent, p.q_dtype, row0, n2, R, p.C, p.Cp, keys, stage_of(wave), sidx);
            team_sync();
            SPFA_STAMP(2);
            if (stage < 2) {
                for (int r = sidx; r < R; r += SW) {
                    const int len = select_row(ctx, r, cbits + (size_t)sidx * (p.Cp >> 5), page_len);
                    if (lane == 0) row_lens[r] = len;
                }
            }
            team_sync();
            if (sidx != 0) return;   // the leader finishes the item alone
            int len_max = 0;
            for (int r = 0; r < R; ++r) len_max = max(len_max, row_lens[r]);
            if (p.max_page_position_length && ctx.split == 0 && lane < kLenPad && len_max > 0)   // cent_select.cpp:124-152
                atomicMax(p.max_page_position_length + (long long)b * kLenPad + lane,
                          (unsigned long long)len_max * (unsigned long long)p.block_size);
            SPFA_STAMP(4);
            // union of the pass's page lists, ascending, with a per-page head mask
            for (int w = 0; w < ((page_len + 63) >> 6); ++w) {
                unsigned long long u = 0;
                unsigned mine = 0;
                for (int r = 0; r < R; ++r) {
                    const unsigned long long x = pbits_all[(size_t)r * n_pwords + w];
                    u |= x;
                    mine |= (unsigned)((x >> lane) & 1ull) << r;
                }
                if ((u >> lane) & 1ull) {
                    const int i = w * 64 + lane;
                    const int e = nU + __popcll(u & lanes_below(lane));
                    if (e < p.max_entries) {
                        PageEntry pe;
                        pe.phys = min(max(bt_row[i], 0), p.n_blocks - 1);
                        pe.valid = (unsigned short)min(p.block_size, seq - i * p.block_size);
                        pe.mask = (unsigned short)mine;
                        list[e] = pe;
                    }
                }
                nU += __popcll(u);
            }
        } else {
            if (sidx != 0) return;
            // list = leading non-sentinel slots of the row, bounded by actual_seq_lengths
            // (…Us2.h:256-294; the kernel stops at the first 0x7fffffff, :308-311)
            const int prow = (p.pos_rows == p.Hq) ? head0 : n2;
            const int32_t* pos = p.block_position + ((long long)b * p.pos_rows + prow) * p.P;
            long long n_tok = p.act_seq ? p.act_seq[b]
                                        : (p.act_seq_uniform >= 0 ? p.act_seq_uniform : (long long)p.P * p.block_size);
            n_tok = max(0ll, min(n_tok, (long long)p.P * p.block_size));
            const int n_slots = (int)((n_tok + p.block_size - 1) / p.block_size);
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
                    pe.valid = (unsigned short)min((long long)p.block_size, n_tok - (long long)j * p.block_size);
                    pe.mask = (unsigned short)((1u << R) - 1u);
                    list[nU + lane] = pe;
                }
                nU += first;
                open = (first == 64);
            }
        }
        SPFA_STAMP(5);
        // pre-scaled q rows (int8 KV: times the per-channel key scale, …Us2.h:2647); the MFMA streamer reads q itself
        for (int t = lane; !MFMA && t < nq * D; t += 64) {
            const int g = t / D, d = t - g * D;
            float v = 0.f;
            if (g < R) {
                v = load_qk_elem(p.q, p.q_dtype, (row0 + g) * D + d) * qscale;
                if (QUANT) v *= load_qk_elem(p.antiq_scale, p.q_dtype, p.antiq_per_tensor ? 0 : (long long)n2 * D + d);
            }
            qst[t] = v;
        }
        SPFA_STAMP(6);
        const int nxt = dynamic ? fetch_end(raw_next, p.counters, cls, total, (int)gridDim.x, lane) : -1;
        SPFA_STAMP(1);
        nU = min(nU, p.max_entries);
        int pc = 0, nch = 0;
        if (p.tail_on && nxt < 0 && nU > 0) {
            // last item: publish the list (and the q rows of the VALU streamer) and offer it in chunks
            // a claim drains the wave's loads in flight: keep the chunks of the wide (MFMA) streamer a few pages long
            pc = tail_geometry(nU, p.tail_chunks, MFMA ? SPFA_TAIL_MIN_PAGES : 1, nch);
            char* bd = board_data((int)blockIdx.x);
            unsigned long long* glist = (unsigned long long*)(bd + tail_list_off);
            wave_lds_sync();
            for (int i = lane; i < nU; i += 64) st_agent(glist + i, ((const unsigned long long*)list)[i]);
            if (!MFMA) {
                float* gq = (float*)(bd + tail_qst_off);
                for (int t = lane; t < nq * D; t += 64) st_agent(gq + t, qst[t]);
            }
            if (lane < 8) {
                const int vals[8] = {nU, pc, nch, n2, (int)row0, R, b, 0};   // TailHdr
                int mine = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) mine = (lane == i) ? vals[i] : mine;
                st_agent((int*)bd + lane, mine);
            }
            if (lane == 0) team[3] = 0;
            wait_stores();   // the data is in memory before the offer
            if (lane == 0) atomicExch(&my_board->state, (unsigned)nch << 16);
        }
        if (lane == 0) {
            const int share = (nU + NS - 1) / NS;
            const int lo = min(nU, ctx.split * share);
            hdr->item = item;
            hdr->nU = min(nU, lo + share) - lo;
            hdr->lo = lo;
            hdr->b = b;
            hdr->n2 = n2;
            hdr->row0 = (int)row0;
            hdr->R = R;
            hdr->next = nxt;
            hdr->pc = pc;
            hdr->nch = nch;
        }
    };

    // ---------------- merge the 4 streamer states of a finished item and store its output rows ----------------
    auto store_out = [&](long long row, int d, int n2, float acc, float lt) {
        float r = lt > 0.f ? acc / lt : 0.f;
        if (QUANT && lt > 0.f) {   // (P.V / l + offset_v) * scale_v, …Us2.h:4160-4174
            // value params follow the key params (…Us2.h:2035-2043)
            const long long pi = p.antiq_per_tensor ? 1 : (long long)p.Hkv * D + (long long)n2 * D + d;
            if (p.antiq_offset) r += load_qk_elem(p.antiq_offset, p.q_dtype, pi);
            r *= load_qk_elem(p.antiq_scale, p.q_dtype, pi);
        }
        const long long oi = row * D + d;
        if (p.out_dtype == kI8) {
            // int8 output (…Us2.h:4016-4111): out * scale2 + offset2 in fp32 (two roundings), then f32 -> f16 -> int8,
            // both steps rounding to nearest with ties away from zero (CAST_ROUND) and saturating.
            const long long qi = p.quant2_per_channel ? (row % p.Hq) * D + d : 0;
            auto param = [&](const void* t) {
                return p.quant2_dtype == kBF16 ? (float)((const __bf16*)t)[qi] : ((const float*)t)[qi];
            };
            float x = __fmul_rn(r, param(p.quant_scale2));
            if (p.quant_offset2) x = __fadd_rn(x, param(p.quant_offset2));
            const float ax = fminf(fabsf(x), 65504.f);
            // f16 below |x| (round toward zero), then up by one f16 step when the remainder reaches half a step
            uint16_t hb = __builtin_bit_cast(uint16_t, (_Float16)ax);
            if ((float)__builtin_bit_cast(_Float16, hb) > ax) --hb;
            const float lo = (float)__builtin_bit_cast(_Float16, hb);
            const float hi = (float)__builtin_bit_cast(_Float16, (uint16_t)(hb + 1));
            const float hx = (hb < 0x7bff && (ax - lo) * 2.f >= (hi - lo)) ? hi : lo;
            int q8 = (int)floorf(hx + 0.5f);
            q8 = !(x == x) ? 0 : (x < 0.f ? -q8 : q8);
            q8 = max(-128, min(127, q8));
            ((int8_t*)p.out)[oi] = (int8_t)q8;
        } else if (p.out_dtype == kF32) ((float*)p.out)[oi] = r;
        else if (p.out_dtype == kBF16) ((__bf16*)p.out)[oi] = (__bf16)r;   // round to nearest even (…Us2.h:3476-3477)
        else ((_Float16*)p.out)[oi] = (_Float16)r;
    };
    // dst == nullptr: normalise and store the output rows; else leave the merged, un-normalised state [nq][D + 2] there
    auto finalize = [&](int slot, int row0, int R, int n2, float* dst_state) {
        const float* comb = combs + (size_t)slot * kStreamWaves * nq * cstride;
        for (int t = threadIdx.x; t < R * D; t += kStreamWaves * 64) {
            const int g = t / D, d = t - g * D;
            float mt = kNegBig;
            for (int w = 0; w < kStreamWaves; ++w) mt = fmaxf(mt, comb[((size_t)w * nq + g) * cstride + D]);
            float acc = 0.f, lt = 0.f;
            for (int w = 0; w < kStreamWaves; ++w) {
                const float* src = comb + ((size_t)w * nq + g) * cstride;
                const float c = fast_exp2(src[D] - mt);
                acc += src[d] * c;
                lt += src[D + 1] * c;
            }
            if (dst_state == nullptr) {
                store_out((long long)row0 + g, d, n2, acc, lt);
            } else {
                float* dst = dst_state + (size_t)g * cstride;
                st_agent(dst + d, acc);
                if (d == 0) { st_agent(dst + D, mt); st_agent(dst + D + 1, lt); }
            }
        }
    };
    // merge the un-normalised states [..][nq][D + 2] whose bit is set in `which` and store the output rows (all waves
    // take part)
    auto merge_states = [&](const float* parts, unsigned long long which, long long row0, int R, int n2) {
        for (int t = threadIdx.x; t < R * D; t += WAVES * 64) {
            const int g = t / D, d = t - g * D;
            float mt = kNegBig;
            for (unsigned long long w = which; w; w &= w - 1) {
                const int u = __ffsll((long long)w) - 1;
                mt = fmaxf(mt, ld_agent(parts + ((size_t)u * nq + g) * cstride + D));
            }
            float acc = 0.f, lt = 0.f;
            for (unsigned long long w = which; w; w &= w - 1) {
                const int u = __ffsll((long long)w) - 1;
                const float* src = parts + ((size_t)u * nq + g) * cstride;
                const float cf = fast_exp2(ld_agent(src + D) - mt);
                acc += ld_agent(src + d) * cf;
                lt += ld_agent(src + D + 1) * cf;
            }
            store_out(row0 + g, d, n2, acc, lt);
        }
    };
    // where the state of item `slot` goes: the output (nullptr), this unit's split slot, or slot 0 of the own board
    auto state_target = [&](bool shared) -> float* {
        if (NS > 1) return p.split_partials + (size_t)blockIdx.x * nq * cstride;
        if (shared) return (float*)(board_data((int)blockIdx.x) + tail_part_off);
        return nullptr;
    };
    auto run_stream = [&](const PageEntry* list, int n_base, const float* qst, int slot, PageCursor& cur, int n2,
                          long long row0, int R) {
        if constexpr (MFMA)
            stream_mfma<KVT, LPT>(p, list, combs + (size_t)slot * kStreamWaves * nq * cstride,
                                  smem + p.off_vst + (size_t)wave * (32 * (D * 2 + 32)), cur, n2, row0, R, nq, qscale,
                                  lane, wave);
        else
            stream_valu<KVT, NQ, LPT>(p, list, qst, combs + (size_t)slot * kStreamWaves * nq * cstride, cur, n2, row0,
                                      n_base, lane,
                                      wave);
        if (cur.mode != 0 && lane == 0 && cur.chunks > 0) {
            atomicAdd(&team[3], cur.chunks);
            atomicMin(&team[1], cur.first);
        }
    };

    // ---------------- item pipeline ----------------
    // First item: static (no counter round trip) and scored by all five waves together, because nothing can
    // be streamed before the first page list exists.
    int slot = 0;
    if (threadIdx.x == 0) team[0] = 0;
    __syncthreads();
    if (wave == 0) SPFA_STAMP(11);
    {
        const int item0 = (int)blockIdx.x;   // grid <= total
        int stage0 = 1;
        unsigned raw0 = kNoFetch;
        if (wave == kStreamWaves && total > (int)gridDim.x)
            raw0 = fetch_begin(p.counters, (xcd + p.class_step) & 7, total, (int)gridDim.x, lane);
        if (p.mode == 1) {
            const ItemCtx c0 = decode(item0);
            // Page-list split: the NS workgroups of an item all need the very same rows.  Each scores every NS-th
            // tile group into global memory and takes the full rows back once the others have delivered (a bounded
            // wait; a workgroup that gives up scores its rows alone).  Sharing a pass between DIFFERENT items of
            // one kv head in large launches was measured too: the hand-off costs more than the 31 us pass.
            bool have_keys = false;
            if (p.pg_on && NS > 1) {
                const int gid = item0 / NS;
                uint32_t* gkeys = p.pg_keys + (size_t)gid * 16 * p.Cp;
                score_rows_fixed<WAVES, DC, QDT>(p.q, p.cent, p.q_dtype, c0.row0, c0.n2, c0.R, p.C, p.Cp, gkeys, stage_of(wave), wave, c0.R, 0,
                                  c0.split, NS);
                wait_stores();
                __syncthreads();
                if (threadIdx.x == 0) {
                    atomicAdd(p.pg_done + gid, 1u);
                    int spins = 0;
                    while (ld_agent(p.pg_done + gid) < (unsigned)NS && spins < (1 << 16)) {
                        __builtin_amdgcn_s_sleep(8);
                        ++spins;
                    }
                    team[2] = ld_agent(p.pg_done + gid) >= (unsigned)NS ? 1 : 0;
                }
                __syncthreads();
                if (team[2]) {
                    for (int t = threadIdx.x; t < c0.R * p.Cp; t += WAVES * 64) keys[t] = ld_agent(gkeys + t);
                    have_keys = true;
                }
                __syncthreads();
            }
            if (!have_keys) {
                score_rows_fixed<WAVES, DC, QDT>(p.q, p.cent, p.q_dtype, c0.row0, c0.n2, c0.R, p.C, p.Cp, keys, stage_of(wave), wave);
                __syncthreads();
            }
            if (wave == 0) SPFA_STAMP(12);
            if (c0.R > 1) {   // one wave per row instead of the selector walking all of them
                const int page_len = page_len_of(c0.b);
                for (int r = wave; r < c0.R; r += WAVES) {
                    const int len = select_row(c0, r, cbits + (size_t)wave * (p.Cp >> 5), page_len);
                    if (lane == 0) row_lens[r] = len;
                }
                __syncthreads();
                if (wave == 0) SPFA_STAMP(13);
                stage0 = 2;
            }
        }
        if (selector) build(0, item0, stage0, raw0);
    }
    __syncthreads();
    if (wave == 0) SPFA_STAMP(14);
#ifdef SPFA_STAMPS
    stamp_item = 1;
#endif
    int prev_row0 = 0, prev_R = 0, prev_n2 = 0;
    bool have_prev = false, prev_shared = false;
#if SPFA_PRIO
    // the streamers' issue slots come first; the selector wave works in the gaps they leave
    if (!selector) __builtin_amdgcn_s_setprio(SPFA_PRIO);
#endif
    while (true) {
        const ItemHdr* h = hdr_at(slot);
        const int item = __builtin_amdgcn_readfirstlane(h->item);
        const int nU = __builtin_amdgcn_readfirstlane(h->nU);
        const int lo = __builtin_amdgcn_readfirstlane(h->lo);
        const int n2 = __builtin_amdgcn_readfirstlane(h->n2);
        const int row0 = __builtin_amdgcn_readfirstlane(h->row0);
        const int R = __builtin_amdgcn_readfirstlane(h->R);
        const int next = __builtin_amdgcn_readfirstlane(h->next);
        const int pc = __builtin_amdgcn_readfirstlane(h->pc);
        const int nch = __builtin_amdgcn_readfirstlane(h->nch);
        if (have_prev && !selector) finalize(slot ^ 1, prev_row0, prev_R, prev_n2, state_target(prev_shared));
        if (item < 0) break;
        if (selector) build(slot ^ 1, next, 0, kNoFetch);
        else {
            if (wave == 0) SPFA_STAMP(8);
            PageCursor cur;
            cur.init(nch > 0 ? 1 : 0, &my_board->state, pc, nch, nU, wave, kStreamWaves);
            run_stream(lists + (size_t)slot * p.max_entries + lo, lo, qsts + (size_t)slot * nq * D, slot, cur, n2, row0, R);
            if (wave == 0) SPFA_STAMP(9);
        }
        __syncthreads();
        if (wave == 0) SPFA_STAMP(10);
#ifdef SPFA_STAMPS
        stamp_item = min(stamp_item + 1, 15);
#endif
        prev_row0 = row0; prev_R = R; prev_n2 = n2; have_prev = true; prev_shared = nch > 0;
        slot ^= 1;
    }
    if (NS > 1) {
        // One unit per workgroup (grid == units), its state is in global memory (agent-scope stores, waited for), count
        // arrivals per item, and let the last workgroup of an item merge the NS states into the output.
        wait_stores();
        __syncthreads();
        const int item = (int)blockIdx.x / NS;
        if (threadIdx.x == 0) team[2] = (int)atomicAdd(p.split_done + item, 1u);
        __syncthreads();
        if (team[2] == NS - 1) {
            const ItemCtx c = decode((int)blockIdx.x);
            merge_states(p.split_partials + (size_t)item * NS * nq * cstride, (1ull << NS) - 1ull, c.row0, c.R, c.n2);
        }
        return;
    }
    if (!p.tail_on) return;
#ifdef SPFA_STAMPS
    stamp_item = 15;
    int dbg_passes = 0, dbg_chunks = 0;
    if (wave == 0) SPFA_STAMP(0);
#endif
    // ---------------- the shared last item: arrive, then help the others ----------------
    // A party adds its chunk count to the board's counter after its state is in memory; the owner adds one more,
    // so the count is complete (chunks + 1) only when the owner has arrived too.  Whoever completes it merges the
    // states.  States: slot 0 the owner's, slot 1 + c the helper party's whose lowest chunk was c.
    auto arrive = [&](int v, int add, int slot_bit, const TailHdr& th) {
        wait_stores();
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long inc = (unsigned long long)add | (slot_bit >= 0 ? (1ull << (32 + slot_bit)) : 0ull);
            const unsigned long long now = atomicAdd(&boards[v].done, inc) + inc;
            const bool complete = (unsigned)(now & 0xffffffffull) == (unsigned)th.nch + 1u;
            team[2] = complete ? 1 : 0;
            team[1] = (int)(now >> 32);
        }
        __syncthreads();
        if (team[2]) {
            const unsigned long long which = 1ull | ((unsigned long long)(unsigned)team[1] << 1);
            merge_states((const float*)(board_data(v) + tail_part_off), which, th.row0, th.R, th.n2);
        }
        __syncthreads();
    };
    if (prev_shared) {
        TailHdr th;
        th.nch = __builtin_amdgcn_readfirstlane(hdr_at(slot ^ 1)->nch);
        th.row0 = prev_row0; th.R = prev_R; th.n2 = prev_n2;
        const int own = team[3];
        arrive((int)blockIdx.x, own + 1, -1, th);
    }
#ifdef SPFA_STAMPS
    if (wave == 0) SPFA_STAMP(1);
#endif
    int empty_scans = 0;
    for (int pass = 0; pass < 2 * (int)gridDim.x; ++pass) {
#ifdef SPFA_STAMPS
        dbg_passes = pass + 1;
#endif
        if (wave == 0) {
            // Look at 64 boards (a different window per workgroup and pass; the whole grid once nothing turns up) and
            // take one of those with the most unclaimed chunks, ties and near-ties broken per workgroup so that the
            // helpers spread out instead of all descending on one board.
            unsigned best = 0;
            const unsigned salt = (blockIdx.x * 0x9E3779B1u) ^ ((unsigned)pass * 0x85EBCA6Bu);
            const int rounds = empty_scans == 0 ? 1 : ((int)gridDim.x + 63) / 64;
            for (int r = 0; r < rounds; ++r) {
                const unsigned i = empty_scans == 0 ? (salt >> 8) + (unsigned)lane * 7u + 1u : (unsigned)(r * 64 + lane + 1);
                const unsigned v = (blockIdx.x + i) % gridDim.x;
                if (v != blockIdx.x) {
                    const unsigned st = ld_agent(&boards[v].state);
                    const int rem = (int)(st >> 16) - (int)(st & 0xffffu);
                    if (rem > 0 && (st >> 16) <= (unsigned)kTailChunksMax)
                        best = max(best, ((unsigned)((rem + 3) >> 2) << 26) | (((v * 0x9E3779B1u) ^ salt) >> 22 << 16) | v);
                }
            }
            best = wave_max_u32(best);
            if (lane == 0) {
                team[2] = best ? (int)(best & 0xffffu) : -1;
                team[3] = 0;
                team[1] = 1 << 20;
            }
        }
        __syncthreads();
        const int v = team[2];
        if (v < 0) {
            if (empty_scans++ > 0) break;   // nothing in the window, then nothing anywhere
            __syncthreads();
            continue;
        }
        empty_scans = 0;
        // the board's data was in memory before it was offered: copy it into this workgroup's LDS (slot 0)
        const char* bd = board_data(v);
        TailHdr th;
        th.nU = ld_agent((const int*)bd + 0); th.pc = ld_agent((const int*)bd + 1); th.nch = ld_agent((const int*)bd + 2);
        th.n2 = ld_agent((const int*)bd + 3); th.row0 = ld_agent((const int*)bd + 4); th.R = ld_agent((const int*)bd + 5);
        th.b = 0; th.pad = 0;
        th.nU = min(max(th.nU, 0), p.max_entries);
        for (int i = threadIdx.x; i < th.nU; i += WAVES * 64)
            ((unsigned long long*)lists)[i] = ld_agent((const unsigned long long*)(bd + tail_list_off) + i);
        if (!MFMA)
            for (int t = threadIdx.x; t < nq * D; t += WAVES * 64) qsts[t] = ld_agent((const float*)(bd + tail_qst_off) + t);
        __syncthreads();   // the copies are complete, and everybody has read team[2] before it is reused
        if (!selector) {
            PageCursor cur;
            cur.init(2, &boards[v].state, th.pc, th.nch, th.nU, wave, kStreamWaves);
            run_stream(lists, 0, qsts, 0, cur, th.n2, th.row0, th.R);
        }
        __syncthreads();
        const int got = team[3];
        const int first = team[1];
#ifdef SPFA_STAMPS
        dbg_chunks += got;
#endif
        if (got > 0) {
            if (!selector)
                finalize(0, th.row0, th.R, th.n2,
                         (float*)(board_data(v) + tail_part_off) + (size_t)(1 + first) * nq * cstride);
            arrive(v, got, first, th);
        }
        __syncthreads();
    }
#ifdef SPFA_STAMPS
    if (wave == 0) SPFA_STAMP(2);
    if (p.debug && threadIdx.x == 0) {
        p.debug[((size_t)blockIdx.x * 16 + 15) * 16 + 3] = (unsigned long long)dbg_passes;
        p.debug[((size_t)blockIdx.x * 16 + 15) * 16 + 4] = (unsigned long long)dbg_chunks;
        p.debug[((size_t)blockIdx.x * 16 + 15) * 16 + 5] = (unsigned long long)(prev_shared ? team[3] + 1 : 0);
    }
#endif
}

template <typename KVT, int NQ, int LPT, bool MFMA>
hipError_t launch_attn_one(const AttnParams& p, hipStream_t s, long long* cap_out) {
    auto kern = attn_kernel<KVT, NQ, LPT, MFMA>;
    if (p.lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)p.lds_bytes);
        if (e != hipSuccess) return e;
    }
    int occ = 0;
    const int n_cu = device_cu_count();
    if (n_cu <= 0) return hipErrorNoDevice;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, attn_threads(MFMA, sizeof(KVT) == 1), p.lds_bytes);
    if (e != hipSuccess) return e;
    int blocks_per_cu = occ < 1 ? 1 : occ;   // residency of the persistent grid
    // 8 streamer waves per CU already saturate HBM; a third workgroup only deepens the memory queues
    if (blocks_per_cu > SPFA_WGS_PER_CU) blocks_per_cu = SPFA_WGS_PER_CU;
    const bool per_head = (p.mode == 0 && p.pos_rows == p.Hq && p.G > 1);
    const int g_eff = per_head ? 1 : p.G;
    const int nq = MFMA ? p.rows_per_pass : NQ;
    const long long total = (long long)p.B * (per_head ? p.Hq : p.Hkv) * ((g_eff + nq - 1) / nq) * p.n_split;
    if (total == 0 && !cap_out) return hipSuccess;
    const long long cap = (long long)n_cu * blocks_per_cu;
    if (cap_out) { *cap_out = cap; return hipSuccess; }   // planning: how many workgroups can be resident
    if (p.n_split > 1 && total > cap) return hipErrorInvalidConfiguration;   // a split needs one workgroup per unit
    const dim3 grid((unsigned)(total < cap ? total : cap));
    static bool said = false;
    if (!said && getenv("SPFA_DEBUG")) {
        said = true;
        fprintf(stderr, "[spfa] attn grid=%u blocks/CU=%d CUs=%d lds=%zu items=%lld mfma=%d\n", grid.x, blocks_per_cu, n_cu,
                p.lds_bytes, total, (int)MFMA);
    }
    hipLaunchKernelGGL(kern, grid, dim3(attn_threads(MFMA, sizeof(KVT) == 1)), p.lds_bytes, s, p);
    return hipGetLastError();
}

// One q head per pass (MHA, or a group whose heads bring their own lists), or int8 KV: VALU streamer.
template <typename KVT, int NQ>
hipError_t launch_attn_valu(const AttnParams& p, int lpt, hipStream_t s, long long* cap_out) {
    switch (lpt) {
#ifndef SPFA_EXPLORE
        case 4: return launch_attn_one<KVT, NQ, 4, false>(p, s, cap_out);
        case 8: return launch_attn_one<KVT, NQ, 8, false>(p, s, cap_out);
        case 32: return launch_attn_one<KVT, NQ, 32, false>(p, s, cap_out);
        case 64:   // 1024-byte head rows: f16 / bf16 at D = 512 (int8 rows end at 512 bytes)
            if constexpr (sizeof(KVT) == 2 && NQ == 1) return launch_attn_one<KVT, NQ, 64, false>(p, s, cap_out);
            break;
#endif
        case 16: return launch_attn_one<KVT, NQ, 16, false>(p, s, cap_out);
    }
    return hipErrorInvalidValue;
}

// Several q heads share the K/V rows (f16 / bf16): MFMA streamer.
template <typename KVT>
hipError_t launch_attn_mfma(const AttnParams& p, int lpt, hipStream_t s, long long* cap_out) {
    switch (lpt) {
#ifndef SPFA_EXPLORE
        case 4: return launch_attn_one<KVT, 16, 4, true>(p, s, cap_out);
        case 8: return launch_attn_one<KVT, 16, 8, true>(p, s, cap_out);
        case 32:
            if constexpr (sizeof(KVT) == 2) return launch_attn_one<KVT, 16, 32, true>(p, s, cap_out);
            break;
#endif
        case 16: return launch_attn_one<KVT, 16, 16, true>(p, s, cap_out);
    }
    return hipErrorInvalidValue;
}

}  // namespace

// Zero two 8-byte aligned ranges (workspace counters, the per-batch maxima that the kernels accumulate with atomicMax)
// in one small launch: unlike a memset node this behaves the same when the stream is being captured into a graph.
__global__ void zero_kernel(unsigned long long* a, size_t na, unsigned long long* b, size_t nb) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (size_t j = i; j < na; j += step) a[j] = 0ull;
    for (size_t j = i; j < nb; j += step) b[j] = 0ull;
}

hipError_t launch_zero(void* a, size_t a_bytes, void* b, size_t b_bytes, hipStream_t s) {
    const size_t na = a ? a_bytes / 8 : 0, nb = b ? b_bytes / 8 : 0;
    if (na + nb == 0) return hipSuccess;
    const size_t most = na > nb ? na : nb;
    const unsigned blocks = (unsigned)((most + 255) / 256 < 64 ? (most + 255) / 256 : 64);
    hipLaunchKernelGGL(zero_kernel, dim3(blocks), dim3(256), 0, s, (unsigned long long*)a, na, (unsigned long long*)b, nb);
    return hipGetLastError();
}

int device_cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return n;
}

// Head dims above 256 would need 128 and more accumulator registers for O^T: those shapes stay on the VALU form
// (int8: 4 q heads per pass; f16 / bf16 at D = 512: one).
bool attention_uses_mfma(int rows_per_pass, int kv_dtype, int D) {
    return rows_per_pass > 1 && D <= 256;
}

int attention_stream_waves(bool mfma, int kv_dtype) { return stream_waves(mfma, kv_dtype == kI8); }
int attention_stage_bytes(bool mfma, int D) {
    return (mfma ? sel_waves(true) : stream_waves(false, false) + sel_waves(false)) * score_stage_bytes(D);
}
int cent_stage_bytes(int D) { return 4 * score_stage_bytes(D); }

int attention_rows_per_pass(int G, int kv_dtype, int D) {
    if (G <= 1) return 1;
    if (kv_dtype != kI8 && D > 256) return 1;   // 1024-byte rows: a wave load is one token row
    if (kv_dtype == kI8 && D > 256) return 4;   // VALU streamer: int8 rows cost 16 accumulators each
    return G < 16 ? G : 16;                     // MFMA streamer: up to 16 columns
}

hipError_t launch_attention(const AttnParams& p, hipStream_t s, long long* cap_out) {
    const int nq = p.rows_per_pass;
    const int esize = (p.kv_dtype == kI8) ? 1 : 2;
    const int lpt = p.D * esize / 16;
    if (attention_uses_mfma(nq, p.kv_dtype, p.D)) {
        if (nq > 16) return hipErrorInvalidValue;
#ifdef SPFA_EXPLORE
        if (p.kv_dtype == kF16) return launch_attn_mfma<_Float16>(p, lpt, s, cap_out);
#ifdef SPFA_EXPLORE_I8
        if (p.kv_dtype == kI8 && lpt == 8) return launch_attn_one<int8_t, 16, 8, true>(p, s, cap_out);
#endif
        return hipErrorInvalidValue;
#else
        switch (p.kv_dtype) {
            case kF16: return launch_attn_mfma<_Float16>(p, lpt, s, cap_out);
            case kBF16: return launch_attn_mfma<__bf16>(p, lpt, s, cap_out);
            case kI8: return launch_attn_mfma<int8_t>(p, lpt, s, cap_out);
        }
        return hipErrorInvalidValue;
#endif
    }
    switch (p.kv_dtype) {
        case kF16: if (nq == 1) return launch_attn_valu<_Float16, 1>(p, lpt, s, cap_out); break;
#ifdef SPFA_EXPLORE_I8
        case kI8: if (nq == 1 && lpt == 8) return launch_attn_one<int8_t, 1, 8, false>(p, s, cap_out); break;
#endif
#ifndef SPFA_EXPLORE
        case kBF16: if (nq == 1) return launch_attn_valu<__bf16, 1>(p, lpt, s, cap_out); break;
        case kI8:
            if (nq == 1) return launch_attn_valu<int8_t, 1>(p, lpt, s, cap_out);
            if (nq == 4 && lpt == 32) return launch_attn_one<int8_t, 4, 32, false>(p, s, cap_out);
            break;
#endif
    }
    return hipErrorInvalidValue;
}

static hipError_t launch_cent(const CentParams& p, int mode, hipStream_t s) {
    const dim3 grid((unsigned)((long long)((p.B + p.seq_per_group - 1) / p.seq_per_group) * p.Hkv));
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
