// fused_rock.hip — the fused multi-step launches of RockSample (the kernels bench.py times).
// Part of libpomdp_hip.so; built by gym_pomdp_amd/_native.py (hipcc --offload-arch=gfx950 -O3 -std=c++17 -c, one object per file).
#include "fused_impl.hip.h"
namespace pomdp {
POMDP_FUSED_LAUNCHER(, Rock1)
POMDP_FUSED_LAUNCHER(, Rock2)
}
