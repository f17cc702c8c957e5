// Attention kernels for i8 KV caches (see spfa_attn_tu.inc).
#if !defined(SPFA_EXPLORE) || defined(SPFA_EXPLORE_I8)
#include "spfa_attn_tu.inc"
namespace spfa {
hipError_t launch_attention_i8(const AttnParams& p, hipStream_t s, long long* cap_out) {
    return launch_attention_t<int8_t>(p, s, cap_out);
}
}  // namespace spfa
#else   // exploration builds leave this type out
#include "spfa_params.h"
namespace spfa {
hipError_t launch_attention_i8(const AttnParams&, hipStream_t, long long*) { return hipErrorInvalidValue; }
}  // namespace spfa
#endif
