"""Reference (numpy) statement of the simulator's random-word contract.

TEST INFRASTRUCTURE — part of the oracle.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; the product path
(gym_pomdp_amd/) never does.

The contract (also restated in DESIGN.md §RNG and implemented independently in
oracle/pomdp_oracle.c and gym_pomdp_amd/csrc/philox.hip.h):

* generator: Philox4x32-10 (Salmon et al., SC'11), multipliers 0xD2511F53 /
  0xCD9E8D57, Weyl key increments 0x9E3779B9 / 0xBB67AE85;
* key   = (seed & 0xffffffff, seed >> 32);
* ctr   = (lane, t & 0xffffffff, t >> 32, (stream << 24) | block);
* the 32-bit word stream of (seed, lane, t, stream) is block 0's four outputs,
  then block 1's, ... ; env code consumes that stream strictly sequentially,
  with numpy's *legacy* RandomState constructions on top of it:
    double   : a = w0 >> 5, b = w1 >> 6, k = a * 2**26 + b, U = k / 2**53
    randint n: mask = bit-smear(n - 1); draw words until (w & mask) <= n - 1
    binomial(1, p): one double, compared against a captured integer threshold.

RockSample / StochasticRock, Network's step(), Tiger and Tag with one opponent deviate from "strictly sequential" in how the words are laid out (not
in how numpy consumes them): see split_words / rock_reset_words / rock_step_words / network_step_words / tiger_words /
tag_step_words / tag_auto_reset_words below
(split high / low blocks, quad-shared streams).

Streams: 0 np.random draws made inside step(); 1 np.random draws made inside
reset(); 2 / 3 the gym-space RNG (Discrete.sample) inside step() / reset()
(Tiger only — since ABI 13 both read the words of stream 0, tiger_words); 4 the benchmark's synthetic random-action policy; 5 the rollout policy's pick
among the legal actions (word k of the stream of the rollout's first call counter picks step k); 6 BattleShip's
"next board": whenever a board is dealt at call counter t (reset(): from stream 1; an auto-reset: the cached board moves
in), the reference's reset() run on stream 6 of (lane, t) gives the board of the episode after it.
"""
import numpy as np

STREAM_STEP = 0
STREAM_RESET = 1
STREAM_STEP_SPACE = 2
STREAM_RESET_SPACE = 3
STREAM_ACTION = 4
STREAM_ROLLOUT = 5
STREAM_STEP_LO = 7    # Network: the per-lane low parts of step()'s doubles (network_step_words), generated on ties only
STREAM_NEXT = 6       # BattleShip: the board of the episode AFTER the one dealt at (lane, t) (include/pomdp_hip.h: board contract)

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: uint32[..., 4], key: uint32[..., 2] (broadcastable) -> uint32[..., 4]."""
    ctr = np.asarray(ctr, dtype=np.uint64)
    key = np.asarray(key, dtype=np.uint64)
    c0, c1, c2, c3 = (ctr[..., i].copy() for i in range(4))
    k0 = key[..., 0].copy()
    k1 = key[..., 1].copy()
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = (k0 + np.uint64(_W0)) & _MASK32
        k1 = (k1 + np.uint64(_W1)) & _MASK32
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def stream_words(seed, lane, t, stream, nwords):
    """First `nwords` 32-bit words of the (seed, lane, t, stream) stream."""
    nblk = (int(nwords) + 3) // 4
    ctr = np.zeros((nblk, 4), dtype=np.uint64)
    ctr[:, 0] = int(lane) & 0xFFFFFFFF
    ctr[:, 1] = int(t) & 0xFFFFFFFF
    ctr[:, 2] = (int(t) >> 32) & 0xFFFFFFFF
    ctr[:, 3] = (int(stream) << 24) | np.arange(nblk, dtype=np.uint64)
    key = np.array([int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    return philox4x32_10(ctr, key).reshape(-1)[:nwords]


def _block(seed, c0, t, stream, block):
    ctr = np.array([int(c0) & 0xFFFFFFFF, int(t) & 0xFFFFFFFF, (int(t) >> 32) & 0xFFFFFFFF,
                    (int(stream) << 24) | int(block)], dtype=np.uint64)
    key = np.array([int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    return philox4x32_10(ctr, key)


def split_words(seed, lane, t, stream, n_doubles):
    """Per-lane *split layout* of a stream whose draws are all doubles: double j takes its high word from element
    j & 3 of block 2 (j >> 2) and its low word from the same element of block 2 (j >> 2) + 1; the kernels generate
    the odd ("low") blocks only when a high word leaves a comparison undecided.  Returns the 2 * n_doubles words
    numpy consumes, in order.  (Network's STEP stream until ABI 11; network_step_words since.)"""
    out = []
    for j in range(n_doubles):
        hi = _block(seed, lane, t, stream, 2 * (j >> 2))[j & 3]
        lo = _block(seed, lane, t, stream, 2 * (j >> 2) + 1)[j & 3]
        out += [int(hi), int(lo)]
    return np.array(out, dtype=np.uint32)


def network_step_words(seed, lane, t, n_doubles):
    """Network's STEP stream (ABI 12): every draw of step() is a double — one per up machine in index order,
    then one for the action (network.py:94-109) — compared with a threshold, and 16 random bits decide such a comparison
    unless they equal the threshold's top 16 bits (probability 2^-16).  So the TOP 16 bits of double j come from a block
    shared by the four lanes of a quad — counter (lane >> 2, t, STEP, block j >> 1), element lane & 3, upper half of the word
    for even j, lower half for odd j: one Philox block serves two draws of each of four lanes — and the 37 bits below them
    from the lane's own stream STEP_LO — counter (lane, t, STEP_LO, block j >> 1), elements 2 (j & 1) and 2 (j & 1) + 1 —
    which the kernels generate on a tie only.  numpy builds the double from two words (a >> 5, b >> 6):
        a_j = Q_j << 16 | X_j >> 16,   b_j = Y_j.
    Returns the 2 * n_doubles words numpy consumes, in order."""
    out = []
    for j in range(n_doubles):
        w = int(_block(seed, lane >> 2, t, STREAM_STEP, j >> 1)[lane & 3])
        q = (w >> 16) if (j & 1) == 0 else (w & 0xFFFF)
        lo = _block(seed, lane, t, STREAM_STEP_LO, j >> 1)
        x, y = int(lo[2 * (j & 1)]), int(lo[2 * (j & 1) + 1])
        out += [(q << 16) | (x >> 16), y]
    return np.array(out, dtype=np.uint32)


def rotr32(w, r):
    w, r = int(w) & 0xFFFFFFFF, int(r) & 31
    return ((w >> r) | (w << (32 - r))) & 0xFFFFFFFF


def rock_reset_words(seed, lane, t, n_rocks, auto_step_block=None):
    """RockSample's RESET stream ("rotated split layout", shared by the four lanes of a quad like the
    STEP stream).  reset() draws one double per rock and only uses sign(U - .5), i.e. the top bit of the double's high
    word (the rest matters only when the top 27 bits are exactly 2^26: a tie, probability 2^-27).  ONE 32-bit word
    therefore serves all K <= 16 rocks of a lane: the Philox counter carries lane >> 2, lane L uses element L & 3, and
    rock j takes as its high word that element of block 0 rotated right by 2 j + 2 bits (j = 15: unrotated) and as its
    low word the same element of block 1 under the same rotation.  The top bits of the rotations are bits 1, 3, ..., 31
    of the element — independent fair bits — so the rock statuses are independent Bernoulli(1/2) exactly as in the
    reference, and they sit where the packed state keeps the upper bit of each rock's 2-bit code.  The kernels
    generate the low block only on a tie.  Returns the 2 * n_rocks words numpy consumes, in order.

    `auto_step_block`: None for a reset() call of its own (stream RESET, blocks 0 and 1).  The reset that follows a done
    step INSIDE that step's call counter (auto-reset) takes the same rotated pair from the step's own SENSOR blocks
    instead — stream STEP, blocks b and b + 1 with b = 0 (RockEnv) or 2 (StochasticRockEnv: block 0 gates the action):
    a step never makes both draws (a CHECK does not end the episode, rock.py:171-175 / 193), so the word is consumed
    exactly once either way and a quad's step costs one Philox block, not two."""
    stream, b = (STREAM_RESET, 0) if auto_step_block is None else (STREAM_STEP, int(auto_step_block))
    hi_w = int(_block(seed, lane >> 2, t, stream, b)[lane & 3])
    lo_w = int(_block(seed, lane >> 2, t, stream, b + 1)[lane & 3])
    out = []
    for j in range(n_rocks):
        rot = (2 * j + 2) & 31
        out += [rotr32(hi_w, rot), rotr32(lo_w, rot)]
    return np.array(out, dtype=np.uint32)


def rock_step_words(seed, lane, t, n_doubles=1):
    """RockSample's STEP stream: split layout, and *shared by the four lanes of a quad* — the Philox counter carries
    lane >> 2 and lane L uses element L & 3, so that one block serves four lanes' sensor draws.  Double j of the step
    (j = 0 for RockEnv's sensor; StochasticRockEnv: j = 0 the action gate, j = 1 the sensor) has its high word in
    block 2 j and its low word in block 2 j + 1."""
    out = []
    for j in range(n_doubles):
        hi = _block(seed, lane >> 2, t, STREAM_STEP, 2 * j)[lane & 3]
        lo = _block(seed, lane >> 2, t, STREAM_STEP, 2 * j + 1)[lane & 3]
        out += [int(hi), int(lo)]
    return np.array(out, dtype=np.uint32)


def tiger_words(seed, lane, t):
    """Tiger (ABI 13): whatever a call with counter t draws — LISTEN's uniform() (tiger.py:140-149), the door a wrong guess
    resamples (state_space.sample(), tiger.py:117-119), the door of the episode that reset() or the auto-reset after a
    right guess starts (tiger.py:60-66) — it reads from the QUAD's STEP stream, like RockSample's sensor: the Philox counter
    carries lane >> 2, lane L uses element L & 3; the double's high word from block 0, its low word (a tie of the top 27 bits
    only: 2^-27) from block 1; a door is bit 0 of the block-0 word (randint(2): mask 1, never rejects).  A call makes at most
    one of these draws that matters (the uniform() a non-LISTEN step draws decides nothing), so one block serves four lanes.
    -> [high word, low word]: the words of np.random for stream STEP, and of the gym-space RNG for STEP_SPACE / RESET_SPACE."""
    return rock_step_words(seed, lane, t, 1)


def tag_step_words(seed, lane, t):
    """Tag with ONE opponent (ABI 13): a step draws only when a TAG fails — the opponent's flight, binomial(1, move_prob) then
    np.random.choice over 2 or 4 admissible moves (tag.py:201-207) — and reads the lane's word W of the QUAD's STEP block 0
    (counter word 0 = lane >> 2, element lane & 3) for both: the double is (W, W') with W' the same element of block 1 (it
    matters on a tie of the top 27 bits only), the choice's randint word is W again (it uses bits 0-1, the double bits 5-31).
    -> the three words np.random consumes."""
    hi, lo = (int(x) for x in rock_step_words(seed, lane, t, 1))
    return np.array([hi, lo, hi], dtype=np.uint32)


def tag_auto_reset_words(seed, lane, t, n_tail=64):
    """Tag with ONE opponent: the reset() that follows a successful TAG inside the step's call (the step itself drew nothing)
    draws randint(29) per cell by masked rejection, 5 bits per attempt (tag.py:43-44, 181-193): attempt i < 6 reads bits
    5 i .. 5 i + 4 of the step's quad word W, later attempts (five of the first six rejected: 4 x 10^-5) the lane's own RESET
    stream from its first word on."""
    w = int(rock_step_words(seed, lane, t, 1)[0])
    head = [(w >> (5 * i)) & 0xFFFFFFFF for i in range(6)]
    return np.concatenate([np.array(head, dtype=np.uint32), stream_words(seed, lane, t, STREAM_RESET, n_tail)])


def synthetic_actions(seed, lane0, n, t, n_actions):
    """The bench's synthetic uniform policy: lanes 4q..4q+3 share block
    ctr=(q, t_lo, t_hi, STREAM_ACTION<<24); action = (word * n_actions) >> 32."""
    lanes = np.arange(lane0, lane0 + n, dtype=np.uint64)
    q = lanes >> np.uint64(2)
    ctr = np.zeros((n, 4), dtype=np.uint64)
    ctr[:, 0] = q & _MASK32
    ctr[:, 1] = int(t) & 0xFFFFFFFF
    ctr[:, 2] = (int(t) >> 32) & 0xFFFFFFFF
    ctr[:, 3] = STREAM_ACTION << 24
    key = np.array([int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    blk = philox4x32_10(ctr, key).astype(np.uint64)
    w = blk[np.arange(n), (lanes & np.uint64(3)).astype(np.int64)]
    return ((w * np.uint64(n_actions)) >> np.uint64(32)).astype(np.int32)


def action_word(seed, lane, t):
    """The synthetic policy's 32-bit word of (seed, lane, t): element lane & 3 of block ctr = (lane >> 2, t, ACTION)."""
    return int(_block(seed, lane >> 2, t, STREAM_ACTION, 0)[lane & 3])


# Known-answer vectors for Philox4x32-10 (Random123 kat_vectors).
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


if __name__ == "__main__":
    for ctr, key, out in KAT:
        got = philox4x32_10(np.array(ctr), np.array(key))
        assert tuple(int(x) for x in got) == out, (ctr, key, [hex(int(x)) for x in got])
    print("philox KAT ok")
