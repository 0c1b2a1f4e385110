// fused_battleship.hip — the fused multi-step launches of BattleShip (one to four mask words).
// Part of libpomdp_hip.so; built by gym_pomdp_amd/_native.py (hipcc --offload-arch=gfx950 -O3 -std=c++17 -c, one object per file).
#include "fused_impl.hip.h"
namespace pomdp {
POMDP_FUSED_LAUNCHER(, BattleShip1)
POMDP_FUSED_LAUNCHER(, BattleShip2)
POMDP_FUSED_LAUNCHER(, BattleShip3)
POMDP_FUSED_LAUNCHER(, BattleShip4)
}
