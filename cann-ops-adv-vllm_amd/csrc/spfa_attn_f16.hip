// Attention kernels for f16 KV caches (see spfa_attn_tu.inc).
#include "spfa_attn_tu.inc"
namespace spfa {
hipError_t launch_attention_f16(const AttnParams& p, hipStream_t s, long long* cap_out) {
    return launch_attention_t<_Float16>(p, s, cap_out);
}
}  // namespace spfa
