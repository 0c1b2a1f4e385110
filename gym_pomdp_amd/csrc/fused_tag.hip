// fused_tag.hip — the fused multi-step launches of Tag.
// Part of libpomdp_hip.so; built by gym_pomdp_amd/_native.py (hipcc --offload-arch=gfx950 -O3 -std=c++17 -c, one object per file).
#include "fused_impl.hip.h"
namespace pomdp {
POMDP_FUSED_LAUNCHER(, TagEnv)
}
