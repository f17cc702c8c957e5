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
// Files: this translation unit holds the standalone ops (scoring, position selection, cluster assignment), the small
// service kernels and the host-side launchers; the attention kernels compile as one translation unit per KV element
// type (spfa_attn_f16.hip, spfa_attn_bf16.hip, spfa_attn_i8.hip, all through spfa_attn_tu.inc).  The device code
// proper lives in spfa_common.inc (helpers), spfa_select.inc (scoring, top-k, page selection), spfa_cent_kernels.inc
// (standalone scoring ops), spfa_assign.inc, spfa_pipeline.inc (work items, queues, sharing of last items),
// spfa_stream_valu.inc / spfa_stream_mfma.inc (the two K/V streamers) and spfa_attn_kernel.inc (the persistent
// attention kernel).
#include "spfa_params.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

namespace spfa {
namespace {

#include "spfa_common.inc"
#include "spfa_select.inc"
#include "spfa_cent_kernels.inc"
#include "spfa_pipeline.inc"   // build knobs and wave counts of the attention kernels (sizes the host reports)
#include "spfa_assign.inc"

}  // namespace

// Zero two 8-byte aligned ranges (workspace counters, the per-batch maxima that the kernels accumulate with atomicMax)
// in one small launch: unlike a memset node this behaves the same when the stream is being captured into a graph.
__global__ void zero_kernel(unsigned long long* a, size_t na, unsigned long long* b, size_t nb) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (size_t j = i; j < na; j += step) a[j] = 0ull;
    for (size_t j = i; j < nb; j += step) b[j] = 0ull;
}

static unsigned long long g_launches = 0;
void count_launch() { __atomic_fetch_add(&g_launches, 1ull, __ATOMIC_RELAXED); }
unsigned long long launches_so_far() { return __atomic_load_n(&g_launches, __ATOMIC_RELAXED); }

hipError_t launch_zero(void* a, size_t a_bytes, void* b, size_t b_bytes, hipStream_t s) {
    const size_t na = a ? a_bytes / 8 : 0, nb = b ? b_bytes / 8 : 0;
    if (na + nb == 0) return hipSuccess;
    const size_t most = na > nb ? na : nb;
    const unsigned blocks = (unsigned)((most + 255) / 256 < 64 ? (most + 255) / 256 : 64);
    count_launch();
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
        count_launch();
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

bool attention_team(const AttnParams& p) {
    return p.sup_items > 1 || (SPFA_TEAM_ALL && p.rows_per_pass == 1 && !p.token_mode && !p.antiq_per_token && !p.narrow);
}
int attention_stream_waves(bool mfma, int kv_dtype, bool narrow) { return stream_waves(mfma, kv_dtype == kI8, narrow); }
int attention_score_waves(bool mfma, int kv_dtype, int D, bool narrow, bool sup) {
    const int all = stream_waves(mfma, kv_dtype == kI8, narrow) + sel_waves(mfma, sup, narrow);
    if (mfma) return all;   // the streamers lend their V tile areas
    return D <= 128 ? all : (all < 5 ? all : 5);   // 8 / 16 KiB tiles: five of them, as much as nine 4 KiB ones
}
int attention_stage_bytes(bool mfma, int kv_dtype, int D, bool narrow, bool sup) {
    return (mfma ? sel_waves(true) : attention_score_waves(false, kv_dtype, D, narrow, sup)) * score_stage_bytes(D);
}
// 4 KiB tiles (D <= 128): every wave; 8 / 16 KiB tiles: four / two waves, the same 33 KiB
int cent_score_waves(int D) { return D <= 128 ? kCentWaves : (D <= 256 ? 4 : 2); }
int cent_stage_bytes(int D) { return cent_score_waves(D) * score_stage_bytes(D); }
int assign_stage_bytes(int D) { return 4 * score_stage_bytes(D); }

int attention_rows_per_pass(int G, int kv_dtype, int D) {
    if (G <= 1) return 1;
    if (kv_dtype != kI8 && D > 256) return 1;   // 1024-byte rows: a wave load is one token row
    if (kv_dtype == kI8 && D > 256) return 4;   // VALU streamer: int8 rows cost 16 accumulators each
    return G < 16 ? G : 16;                     // MFMA streamer: up to 16 columns
}

// One translation unit per KV element type holds the attention kernels (spfa_attn_f16.hip, spfa_attn_bf16.hip,
// spfa_attn_i8.hip).
hipError_t launch_attention_f16(const AttnParams& p, hipStream_t s, long long* cap_out);
hipError_t launch_attention_bf16(const AttnParams& p, hipStream_t s, long long* cap_out);
hipError_t launch_attention_i8(const AttnParams& p, hipStream_t s, long long* cap_out);
hipError_t launch_attention(const AttnParams& p, hipStream_t s, long long* cap_out) {
    switch (p.kv_dtype) {
        case kF16: return launch_attention_f16(p, s, cap_out);
        case kBF16: return launch_attention_bf16(p, s, cap_out);
        case kI8: return launch_attention_i8(p, s, cap_out);
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
    count_launch();
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
    count_launch();
    hipLaunchKernelGGL(assign_kernel, dim3((unsigned)(groups * p.Hkv)), dim3(kThreads), p.lds_bytes, s, p);
    return hipGetLastError();
}

hipError_t launch_select_position(const SelectParams& p, hipStream_t s) {
    const long long rows = (long long)p.B * p.Hq;
    if (rows == 0) return hipSuccess;
    const dim3 grid((unsigned)((rows + 3) / 4));
    count_launch();
    hipLaunchKernelGGL(select_position_kernel, grid, dim3(kThreads), (size_t)4 * p.k * sizeof(int32_t), s, p);
    return hipGetLastError();
}

}  // namespace spfa
