// envs.hip.h — the five env types the generic kernels are instantiated for; the Env interface is described in
// envs_common.hip.h, each env lives in envs/<name>.hip.h.
#pragma once
#include "envs_common.hip.h"
#include "envs/rock.hip.h"
#include "envs/tag.hip.h"
#include "envs/battleship.hip.h"
#include "envs/tiger.hip.h"
#include "envs/network.hip.h"
