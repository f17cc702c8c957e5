// Attention kernels for bf16 KV caches (see spfa_attn_tu.inc).
#if !defined(SPFA_EXPLORE)
#include "spfa_attn_tu.inc"
namespace spfa {
hipError_t launch_attention_bf16(const AttnParams& p, hipStream_t s, long long* cap_out) {
    return launch_attention_t<__bf16>(p, s, cap_out);
}
}  // namespace spfa
#else   // exploration builds leave this type out
#include "spfa_params.h"
namespace spfa {
hipError_t launch_attention_bf16(const AttnParams&, hipStream_t, long long*) { return hipErrorInvalidValue; }
}  // namespace spfa
#endif
