// gfx950 (MI355X / CDNA4) kernels for the sparse paged fusion attention chain:
//   ComputeCent -> CentSelect / SelectPosition -> SparsePagedAttention, and the fused op.
//
// Building blocks (all wave64):
//   * Q.Cent^T   : v_mfma_f32_16x16x32_{f16,bf16}; A = the q heads of one kv head (zero padded to 16
//                  rows), B = 16 centroid rows, read 1 KiB contiguous per instruction and turned into
//                  operand order through a wave-private LDS tile.
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
//
// Files: this one holds the small device helpers and the host-side launchers; the device code proper is included
// from spfa_select.inc (scoring, top-k, page selection), spfa_cent_kernels.inc (standalone scoring ops),
// spfa_pipeline.inc (work items, queues, sharing of last items), spfa_stream_valu.inc / spfa_stream_mfma.inc (the
// two K/V streamers) and spfa_attn_kernel.inc (the persistent attention kernel).
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
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

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

#include "spfa_select.inc"
#include "spfa_cent_kernels.inc"
#include "spfa_pipeline.inc"
#include "spfa_stream_valu.inc"
#include "spfa_stream_mfma.inc"
#include "spfa_attn_kernel.inc"
#include "spfa_assign.inc"

constexpr int kMaxDevices = 64;
// Raise a kernel's dynamic-LDS limit on the current device when `bytes` exceeds what was set before (the limit is a
// property of the kernel on a device, and executors planned with different sizes may be launched in any order).
// `table` holds the limit already set per device.
hipError_t ensure_lds_limit(const void* kern, int* table, size_t bytes) {
    if (bytes <= 48 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    if (__atomic_load_n(&table[dev], __ATOMIC_ACQUIRE) >= (int)bytes) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    int seen = __atomic_load_n(&table[dev], __ATOMIC_RELAXED);
    while (seen < (int)bytes && !__atomic_compare_exchange_n(&table[dev], &seen, (int)bytes, false, __ATOMIC_RELEASE,
                                                              __ATOMIC_RELAXED)) {}
    return hipSuccess;
}

// cap_out != nullptr (phase 1): set the LDS limit, ask the runtime how many workgroups fit a CU, report the resident
// grid.  Otherwise (phase 2): launch; with p.grid_cap from phase 1 this is nothing but the launch itself.
template <typename KVT, int NQ, int LPT, bool MFMA, int VAR = 0>
hipError_t launch_attn_one(const AttnParams& p, hipStream_t s, long long* cap_out) {
    auto kern = attn_kernel<KVT, NQ, LPT, MFMA, VAR>;
    static int lds_limit[kMaxDevices];
    long long cap = p.grid_cap;
    if (cap_out || cap <= 0) {
        hipError_t e = ensure_lds_limit((const void*)kern, lds_limit, p.lds_bytes);
        if (e != hipSuccess) return e;
        int occ = 0;
        const int n_cu = device_cu_count();
        if (n_cu <= 0) return hipErrorNoDevice;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, attn_threads(MFMA, sizeof(KVT) == 1), p.lds_bytes);
        if (e != hipSuccess) return e;
        int blocks_per_cu = occ < 1 ? 1 : occ;   // residency of the persistent grid
        // 8 streamer waves per CU already saturate HBM; a third workgroup only deepens the memory queues
        if (blocks_per_cu > SPFA_WGS_PER_CU) blocks_per_cu = SPFA_WGS_PER_CU;
        cap = (long long)n_cu * blocks_per_cu;
        static bool said = false;
        if (!said && getenv("SPFA_DEBUG")) {
            said = true;
            fprintf(stderr, "[spfa] attn blocks/CU=%d CUs=%d lds=%zu mfma=%d\n", blocks_per_cu, n_cu, p.lds_bytes, (int)MFMA);
        }
    }
    if (cap_out) { *cap_out = cap; return hipSuccess; }   // planning: how many workgroups can be resident
    const bool per_head = (p.mode == 0 && p.pos_rows == p.Hq && p.G > 1);
    const int g_eff = per_head ? 1 : p.G;
    const int nq = MFMA ? p.rows_per_pass : NQ;
    const long long total = (long long)p.B * (per_head ? p.Hq : p.Hkv) * ((g_eff + nq - 1) / nq) * p.n_split;
    if (total == 0) return hipSuccess;
    if (p.n_split > 1 && total > cap) return hipErrorInvalidConfiguration;   // a split needs one workgroup per unit
    const dim3 grid((unsigned)(total < cap ? total : cap));
    // pre-scored items: the kernel takes the grid as the distance between a workgroup's items
    if (p.pre_items > 1 && (total != (long long)p.pre_items * grid.x || grid.x % (unsigned)p.Hkv != 0 || p.mode != 1 || g_eff > nq ||
                            p.pre_items * nq > 16))
        return hipErrorInvalidConfiguration;
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

// Per-batch lengths as kernel arguments: up to kLengthsPerLaunch values per launch (3.5 KiB of the 4 KiB argument space).
constexpr int kLengthsPerLaunch = 448;
struct LengthPack { long long v[kLengthsPerLaunch]; };
__global__ void store_lengths_kernel(long long* dst, const LengthPack pack, int n) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n) dst[i] = pack.v[i];
}
hipError_t launch_store_lengths(long long* dst, const long long* host, int n, hipStream_t s) {
    for (int base = 0; base < n; base += kLengthsPerLaunch) {
        LengthPack pack;
        const int m = n - base < kLengthsPerLaunch ? n - base : kLengthsPerLaunch;
        for (int i = 0; i < m; ++i) pack.v[i] = host[base + i];
        for (int i = m; i < kLengthsPerLaunch; ++i) pack.v[i] = 0;
        hipLaunchKernelGGL(store_lengths_kernel, dim3((m + 255) / 256), dim3(256), 0, s, dst + base, pack, m);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// Scratch for the entry point that must not allocate: every GPU this library runs on holds one copy of this array from
// the moment the code object is loaded until the process ends.  In host code the symbol is only a handle; the device
// address is looked up once per device (with that device current) and kept.
static __device__ __align__(256) __attribute__((used)) unsigned char scratch[kStaticScratchBytes];
hipError_t static_scratch(void** ptr, int* device) {
    static void* addr[kMaxDevices];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    void* a = __atomic_load_n(&addr[dev], __ATOMIC_ACQUIRE);
    if (a == nullptr) {
        e = hipGetSymbolAddress(&a, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (a == nullptr) return hipErrorInvalidSymbol;
        __atomic_store_n(&addr[dev], a, __ATOMIC_RELEASE);
    }
    *ptr = a;
    if (device) *device = dev;
    return hipSuccess;
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
int attention_stage_bytes(bool mfma, int kv_dtype, int D) {
    return (mfma ? sel_waves(true) : stream_waves(false, kv_dtype == kI8) + sel_waves(false)) * score_stage_bytes(D);
}
int cent_stage_bytes(int D) { return kCentWaves * score_stage_bytes(D); }
int assign_stage_bytes(int D) { return 4 * score_stage_bytes(D); }

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
#ifndef SPFA_EXPLORE
    if (p.token_mode) {   // token-level lists: one q head per pass, head dims 64 and 128
        if (nq != 1 || p.mode != 0) return hipErrorInvalidValue;
        switch (p.kv_dtype) {
            case kF16:
                if (lpt == 16) return launch_attn_one<_Float16, 1, 16, false, 1>(p, s, cap_out);
                if (lpt == 8) return launch_attn_one<_Float16, 1, 8, false, 1>(p, s, cap_out);
                break;
            case kBF16:
                if (lpt == 16) return launch_attn_one<__bf16, 1, 16, false, 1>(p, s, cap_out);
                if (lpt == 8) return launch_attn_one<__bf16, 1, 8, false, 1>(p, s, cap_out);
                break;
            case kI8:
                if (lpt == 8) return launch_attn_one<int8_t, 1, 8, false, 1>(p, s, cap_out);
                if (lpt == 4) return launch_attn_one<int8_t, 1, 4, false, 1>(p, s, cap_out);
                break;
        }
        return hipErrorInvalidValue;
    }
    if (p.antiq_per_token) {   // int8 KV with per-token parameters: one q head per pass
        if (nq != 1 || p.kv_dtype != kI8) return hipErrorInvalidValue;
        switch (lpt) {
            case 4: return launch_attn_one<int8_t, 1, 4, false, 2>(p, s, cap_out);
            case 8: return launch_attn_one<int8_t, 1, 8, false, 2>(p, s, cap_out);
            case 16: return launch_attn_one<int8_t, 1, 16, false, 2>(p, s, cap_out);
            case 32: return launch_attn_one<int8_t, 1, 32, false, 2>(p, s, cap_out);
        }
        return hipErrorInvalidValue;
    }
#endif
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
    static int lds_limit[2][kMaxDevices];
    const hipError_t le = ensure_lds_limit(kern, lds_limit[mode], p.lds_bytes);
    if (le != hipSuccess) return le;
    if (mode == 0) hipLaunchKernelGGL(cent_kernel<0>, grid, dim3(kCentWaves * 64), p.lds_bytes, s, p);
    else hipLaunchKernelGGL(cent_kernel<1>, grid, dim3(kCentWaves * 64), p.lds_bytes, s, p);
    return hipGetLastError();
}
hipError_t launch_compute_cent(const CentParams& p, hipStream_t s) { return launch_cent(p, 0, s); }
hipError_t launch_cent_select(const CentParams& p, hipStream_t s) { return launch_cent(p, 1, s); }

size_t assign_lds_bytes(int rows_per_pass, int Cp, int D) {
    return (size_t)rows_per_pass * Cp * 4 + (size_t)16 * D * 2 + 16 * sizeof(AssignSlot) + 16 + (size_t)assign_stage_bytes(D);
}
hipError_t launch_assign(const AssignParams& p, hipStream_t s) {
    const long long groups = ((long long)p.B * p.max_new + p.rows_per_pass - 1) / p.rows_per_pass;
    if (groups <= 0 || p.Hkv <= 0) return hipSuccess;
    if (groups * p.Hkv > 0x7fffffffll) return hipErrorInvalidConfiguration;
    static int lds_limit[kMaxDevices];
    const hipError_t le = ensure_lds_limit((const void*)assign_kernel, lds_limit, p.lds_bytes);
    if (le != hipSuccess) return le;
    hipLaunchKernelGGL(assign_kernel, dim3((unsigned)(groups * p.Hkv)), dim3(kThreads), p.lds_bytes, s, p);
    return hipGetLastError();
}

hipError_t launch_select_position(const SelectParams& p, hipStream_t s) {
    const long long rows = (long long)p.B * p.Hq;
    if (rows == 0) return hipSuccess;
    const dim3 grid((unsigned)((rows + 3) / 4));
    hipLaunchKernelGGL(select_position_kernel, grid, dim3(kThreads), (size_t)4 * p.k * sizeof(int32_t), s, p);
    return hipGetLastError();
}

}  // namespace spfa
