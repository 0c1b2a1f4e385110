// step_other.hip — reset / step launchers of Tag, BattleShip, Tiger and Network.
// Part of libpomdp_hip.so; built by gym_pomdp_amd/_native.py (hipcc --offload-arch=gfx950 -O3 -std=c++17 -c, one object per file).
#include "step_impl.hip.h"
namespace pomdp {
POMDP_STEP_LAUNCHERS(, TagEnv)
POMDP_STEP_LAUNCHERS(, BattleShip1)
POMDP_STEP_LAUNCHERS(, BattleShip2)
POMDP_STEP_LAUNCHERS(, BattleShip3)
POMDP_STEP_LAUNCHERS(, BattleShip4)
POMDP_STEP_LAUNCHERS(, TigerEnv)
POMDP_STEP_LAUNCHERS(, NetworkEnv)
}
