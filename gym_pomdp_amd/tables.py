"""Static configuration tables of the in-scope envs, restated from the reference
(d3sm0/gym_pomdp; paths relative to gym_pomdp/envs/), and the integer Bernoulli
thresholds captured from numpy's legacy binomial (tests/golden/thresholds.json is
the fixture they are checked against; SURVEY.md §9.1 lists the same values).
"""
import math

# rock.py:43-64 — board_size -> (allowed num_rocks, init_pos, rock_pos).  Every listed
# coordinate is stamped on the grid even when num_rocks is smaller (rock.py:110-111), and
# RockSample(15,15) lists 16 coordinates with (1,2) twice: both are reproduced as-is.
ROCK_CONFIG = {
    2: ((2, 1), (0, 0), ((1, 0),)),
    4: ((4, 3), (0, 0), ((1, 0), (3, 1), (2, 3))),
    7: ((7, 8), (0, 3), ((2, 0), (0, 1), (3, 1), (6, 3), (2, 4), (3, 4), (5, 5), (1, 6))),
    11: ((11, 11), (0, 5),
         ((0, 3), (0, 7), (1, 8), (2, 4), (3, 3), (3, 8), (4, 3), (5, 8), (6, 1), (9, 3), (9, 9))),
    15: ((15, 15), (0, 5),
         ((0, 7), (0, 3), (1, 2), (1, 2), (2, 6), (3, 7), (3, 2), (4, 7), (5, 2), (6, 9), (9, 7), (9, 1),
          (11, 8), (13, 10), (14, 9), (12, 2))),
}

# np.random.binomial(1, eff(d)) on U = k / 2**53 returns 1 iff k <= ROCK_THR[d], with
# eff(d) = (1 + 2**(-d/20)) / 2 and d the L1 distance (rock.py:383-387, 401-407).
ROCK_THR = (
    9007199254740992, 8853790118380056, 8705606660380041, 8562470874952118,
    8424210819838096, 8290660409764310, 8161659216931231, 8037052278299120,
    7916689909438254, 7800427524720092, 7688125463633382, 7579648823016592,
    7474867295005110, 7373655010498564, 7275890387960212, 7181455987366794,
    7090238369133370, 7002127957843708, 6917018910622514, 6834808989991382,
    6755399441055744, 6678694872875276, 6604603143875268, 6533035251161307,
    6463905223604296, 6397130018567403, 6332629422150863, 6270325952834808,
    6210144768404375,
)
ROCK_EFF_HEX = (
    "0x1.0000000000000p+0", "0x1.f7479a6ec0218p-1", "0x1.eedb4008bd589p-1", "0x1.e6b859ae6b1b6p-1",
    "0x1.dedc66d6df090p-1", "0x1.d744fccad69d6p-1", "0x1.cfefc5e67299fp-1", "0x1.c8da80e16d9f0p-1",
    "0x1.c203001d9572ep-1", "0x1.bb6728fb505dcp-1", "0x1.b504f333f9de6p-1", "0x1.aeda6839e3c90p-1",
    "0x1.a8e5a29dca9b6p-1", "0x1.a324cd798d804p-1", "0x1.9d9623dffc194p-1", "0x1.9837f0518db8ap-1",
    "0x1.93088c35d733ap-1", "0x1.8e065f5995efcp-1", "0x1.892fdf7128332p-1", "0x1.84838f9f4c1d6p-1",
    "0x1.8000000000000p-1", "0x1.7ba3cd376010cp-1", "0x1.776da0045eac4p-1", "0x1.735c2cd7358dbp-1",
    "0x1.6f6e336b6f848p-1", "0x1.6ba27e656b4ebp-1", "0x1.67f7e2f3394cfp-1", "0x1.646d4070b6cf8p-1",
    "0x1.6101800ecab97p-1",
)
# eff(d) = (1 + 2**(-d/20)) / 2 as the reference's float64 arithmetic produces it (rock.py:383-387); _compute_prob returns it
ROCK_EFF = tuple(float.fromhex(h) for h in ROCK_EFF_HEX)
TAG_MOVE_THR = 7205759403792794          # binomial(1, .8)  -> 1 iff k <= thr   (tag.py:204)
NET_FAIL_THR = 8106479329266893          # binomial(1, .1)  -> 1 iff k >  thr   (network.py:97)
NET_FAIL_NEIGHBOUR_THR = 6034823500676464  # binomial(1, .33) -> 1 iff k >  thr (network.py:99)
NET_OBS_THR = 8556839292003942           # binomial(1, .95) -> 1 iff k <= thr   (network.py:106-109)
TIGER_LISTEN_THR = 7656119366529843      # uniform() > .85 iff k > thr          (tiger.py:141-148)


def bernoulli_threshold(p):
    """Threshold of numpy's legacy binomial(1, p) for a p that has no captured constant
    (only Tag's non-default move_prob needs this): 'le' sense for p > .5, 'gt' for p <= .5.
    Follows distributions.c: inversion with qn = exp(log(q)); host libm, same image as numpy."""
    if p <= 0.5:
        q = 1.0 - p
        return math.floor(math.exp(math.log(q)) * 2 ** 53), "gt"
    q = 1.0 - (1.0 - p)
    return math.floor(math.exp(math.log(q)) * 2 ** 53), "le"


def network_neighbours(n_machines, problem_type):
    """network.py:144-168."""
    nb = [[] for _ in range(n_machines)]
    if problem_type == 3:  # make_3legs_neighbours
        assert n_machines >= 4 and n_machines % 3 == 1
        nb[0] += [1, 2, 3]
        for idx in range(1, n_machines):
            if idx < n_machines - 3:
                nb[idx].append(idx + 3)
            if idx <= 4:
                nb[idx].append(0)
            else:
                nb[idx].append(idx - 3)
    else:  # make_ring_neighbours
        for idx in range(n_machines):
            nb[idx].append((idx + 1) % n_machines)
            nb[idx].append((idx + n_machines - 1) % n_machines)
    return nb
