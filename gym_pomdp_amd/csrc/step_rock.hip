// step_rock.hip — reset / step launchers of the RockSample family (RockSample and StochasticRock, one and two state words).
// Part of libpomdp_hip.so; built by gym_pomdp_amd/_native.py (hipcc --offload-arch=gfx950 -O3 -std=c++17 -c, one object per file).
#include "step_impl.hip.h"
namespace pomdp {
POMDP_STEP_LAUNCHERS(, Rock1)
POMDP_STEP_LAUNCHERS(, Rock2)
POMDP_STEP_LAUNCHERS(, StochRock1)
POMDP_STEP_LAUNCHERS(, StochRock2)
}
