/*
 * pomdp_hip.h — C ABI of libpomdp_hip.so, the MI355X (gfx950) batched step()/reset()
 * path for gym_pomdp's discrete envs.
 *
 * The reference (d3sm0/gym_pomdp) is pure Python and has no FFI; the boundary it
 * exposes for this path is the gym.Env duck type of each env class:
 *     reset() -> ob                     step(a) -> (ob, reward, done, info)
 * Each entry point below replaces the *arithmetic* of one of those methods for a
 * whole batch of independent env instances ("lanes"); the reference method it
 * stands in for is cited per function (paths relative to gym_pomdp/envs/).
 * The modules under gym_pomdp_amd/envs/ are the host-side mirror that calls these through ctypes
 * (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every pointer marked "device" is HBM the caller owns (e.g. a torch tensor's
 *     data_ptr()); the library never allocates or frees device memory, and never synchronises (pomdp_step_sync / pomdp_reset_sync / pomdp_stream_sync,
 *     which exist to do so, aside; those two keep ONE 64-byte block of pinned, portable host memory per calling thread — the flag
 *     a one-lane kernel publishes its outputs through — allocated at the thread's first such call and freed when the thread ends);
 *   - `state` is struct-of-arrays: uint32 [words][n], word-major, lane i at state[w*n + i];
 *   - params structs are read on the host at call time and passed to the kernel
 *     by value (kernarg) — they may live on the caller's stack;
 *   - `stream` is a hipStream_t (NULL = the null stream); launches are asynchronous; the calling thread's
 *     current HIP device must be the one that owns `stream` and the buffers (the library never calls
 *     hipSetDevice);
 *   - lanes are globally numbered: lane = lane0 + i.  Random draws depend only on
 *     (seed, lane, t, stream-id), never on n, the grid or the GPU count, so a batch
 *     sharded over several GPUs reproduces the single-GPU result exactly;
 *   - return value: 0 = ok, > 0 = hipError_t of the launch, < 0 = POMDP_E_*.
 *
 * Random-word contract: Philox4x32-10, key = (seed lo, seed hi),
 * ctr = (lane, t lo, t hi, stream_id << 24 | block); numpy legacy constructions on
 * top (res53 doubles, masked-rejection randint).  RockSample / StochasticRock lay their doubles out
 * "split": the high word of a double in one block, its low word in the following block, generated only when the
 * high word leaves a comparison undecided.  Both RockSample streams are shared by the four lanes of a quad: counter
 * word 0 = lane / 4, lane L reads element L % 4 of every block.  step: block 2j for the step's double j (RockEnv: j = 0
 * the sensor; StochasticRock: j = 0 the action gate, j = 1 the sensor).  reset: rock j reads the lane's element of block 0
 * of stream RESET rotated right by 2j + 2 bits (reset() only uses the top bit of the double: one word serves the lane's
 * sixteen rocks), its low word the element of block 1 under the same rotation.  The reset that follows a done step inside
 * that step's call (POMDP_AUTO_RESET) reads the same rotated pair from the step's own SENSOR blocks — stream STEP, blocks
 * b and b + 1, b = 0 (RockEnv) / 2 (StochasticRock) — instead: a step never makes both draws (a CHECK does not end the
 * episode), so each word is consumed once either way and a quad's step costs one Philox block (ABI 10).
 * Network's step (ABI 12): one double per up machine, then one for the action (network.py:94-109).  16 bits decide a
 * comparison with a threshold unless they equal the threshold's top 16 bits, so the TOP 16 bits of double j are a half —
 * upper for even j, lower for odd j — of element lane % 4 of block j / 2 of the QUAD's stream STEP (counter word 0 =
 * lane / 4): one block serves two draws of each of four lanes.  The 37 bits below them come from the lane's own stream
 * STEP_LO (block j / 2, elements 2 (j % 2) and 2 (j % 2) + 1) and are generated on such a tie (2^-16 per draw) only.
 * Tiger (ABI 13): a call with counter t makes at most one draw that matters — LISTEN's uniform() (tiger.py:140-149), the door
 * a wrong guess resamples (tiger.py:117-119), the door of the episode that reset() or the auto-reset after a right guess
 * starts (tiger.py:60-66) — and reads it from the QUAD's stream STEP (counter word 0 = lane / 4, lane L element L % 4): the
 * double's high word, or the door (bit 0), from block 0; the double's low word (a tie of the top 27 bits only) from block
 * 1.  One block serves four lanes.  (Streams STEP_SPACE / RESET_SPACE — the gym-space RNG's own — are no longer read.)
 * Consequence for callers: pomdp_tiger_reset and pomdp_tiger_step of the same lanes must not be given the same (seed, t) —
 * the step's wrong-door resample would redraw the door reset() just dealt.  A call counter that advances with every call
 * (what the host mirror keeps: reset() is call t, the first step call t + 1) never does; re-seeding with the SAME seed
 * rewinds it, so follow such a seed() by reset(), as the reference's callers do (network.py:176-177 in the other order is
 * harmless there: its reset() ran under the previous seed).
 * Tag with ONE opponent (ABI 13): a step draws only when a TAG fails (the opponent's flight: binomial(1, move_prob), then
 * np.random.choice over 2 or 4 moves, tag.py:201-207) or succeeds (the reset that follows inside the call: randint(29) per
 * cell, tag.py:181-193) — never both — and reads the lane's word W of the QUAD's STEP block 0 for either: the flight's double
 * is (W, the same element of block 1 — a tie of the top 27 bits only), its choice word is W again (bits 0-1; the double uses
 * bits 5-31); the auto-reset's attempt i < 6 reads bits 5 i .. 5 i + 4 of W, later attempts (4 x 10^-5 of the resets) the
 * lane's own stream RESET from its first word on.  pomdp_tag_reset itself and games with more opponents keep the
 * sequential per-lane streams.
 */
#ifndef POMDP_HIP_H
#define POMDP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POMDP_ABI_VERSION 14

enum {
    POMDP_E_BADARG = -1,     /* NULL pointer, n < 0, n + lane0 > 2^32 */
    POMDP_E_BADPARAMS = -2,  /* params outside what the packed layout supports */
};

/* flags for *_step */
enum {
    POMDP_AUTO_RESET = 1,    /* done lanes get a fresh episode in the same call (stream RESET of the same t; RockSample:
                                the step's own sensor blocks, BattleShip: the cached board — see the contracts above and
                                below); without it `done` is in/out and done lanes freeze: (ob, reward, done) = (0, 0, 1) */
    POMDP_FUSE_STEPS = 2,    /* pomdp_rollout_synthetic only: consecutive steps may share a launch (see there) */
};

/* id of the word streams of one (seed, lane, t) */
enum { POMDP_STREAM_STEP = 0, POMDP_STREAM_RESET = 1, POMDP_STREAM_STEP_SPACE = 2,
       POMDP_STREAM_RESET_SPACE = 3, POMDP_STREAM_ACTION = 4, POMDP_STREAM_ROLLOUT = 5, POMDP_STREAM_NEXT = 6,
       POMDP_STREAM_STEP_LO = 7 };

/* env kinds for the generic entry points */
enum { POMDP_ENV_ROCK = 0, POMDP_ENV_TAG = 1, POMDP_ENV_BATTLESHIP = 2, POMDP_ENV_TIGER = 3, POMDP_ENV_NETWORK = 4 };

/* ---- RockSample  (rock.py:96-407) ---------------------------------------- */
/* state words: 1 if num_rocks <= 12 else 2.  64-bit view s = w0 | w1 << 32:
 * bits 0-3 x, 4-7 y, bits 8+2j..9+2j = Rock j status + 1 (0 bad, 1 collected, 2 good). */
typedef struct pomdp_rock_params {
    int32_t  size;          /* board is size x size                       rock.py:108 */
    int32_t  num_rocks;     /* K <= 16; actions = 5 + K                   rock.py:113 */
    int32_t  start_x, start_y;                                         /* rock.py:107 */
    int8_t   rock_x[16], rock_y[16];                                   /* rock.py:106 */
    int8_t   grid[256];     /* grid[x * 16 + y] = rock id stamped at (x, y) or -1    rock.py:110-111 */
    uint64_t thr[32];       /* thr[d]: sensor correct iff k53 <= thr[d], d = L1 distance  rock.py:383-407 */
    double   eff[32];       /* eff(d) itself, returned by pomdp_compute_prob            rock.py:383-387 */
    int32_t  stochastic;    /* 1 = StochasticRockEnv (rock.py:428-504): the action is applied only when a
                               binomial(1, p_move) draw succeeds, penalties are 0 and do not terminate */
    int32_t  act_gt;        /* stochastic, 1: act iff k53 > act_thr instead — numpy's binomial(1, p_move) for p_move <= .5 (ABI 14) */
    uint64_t act_thr;       /* stochastic: act iff k53 <= act_thr (first double of stream STEP)  rock.py:443 */
} pomdp_rock_params;

/* replaces RockEnv.reset (rock.py:236-241, 266-271, 78-86).  ob (device, may be NULL) <- 0. */
int pomdp_rock_reset(const pomdp_rock_params *p, uint32_t *state, int32_t *ob, int64_t n,
                     uint64_t seed, uint32_t lane0, uint64_t t, void *stream);
/* replaces RockEnv.step (rock.py:123-194).  reward in {-100,-10,0,10}; err (device uint32, may be
 * NULL) counts lanes whose action was outside [0, 5+K): they are left untouched, (ob,reward,done)=(0,0,0). */
int pomdp_rock_step(const pomdp_rock_params *p, uint32_t *state, const int32_t *action, int32_t *ob,
                    int32_t *reward, uint8_t *done, uint32_t *err, int64_t n,
                    uint64_t seed, uint32_t lane0, uint64_t t, int flags, void *stream);

/* ---- Tag  (tag.py:36-291) -------------------------------------------------- */
/* state: 1 word: bits 0-4 agent cell, bits 5+5j.. opponent j cell (j < 4), bits 25-31 num_opp
 * (7-bit two's complement, saturating at -64). */
typedef struct pomdp_tag_params {
    int32_t  num_opponents; /* 1..4                                       tag.py:87 */
    int32_t  obs_cells;     /* "opponent seen" observation value          tag.py:94 */
    uint64_t move_thr;      /* opponent moves iff k53 <= move_thr         tag.py:204 */
    int32_t  move_gt;       /* 1: ... iff k53 > move_thr instead — numpy's binomial(1, p) for p <= .5 (ABI 12) */
    int32_t  reserved;
} pomdp_tag_params;

/* replaces TagEnv.reset (tag.py:97-102, 181-193) */
int pomdp_tag_reset(const pomdp_tag_params *p, uint32_t *state, int32_t *ob, int64_t n,
                    uint64_t seed, uint32_t lane0, uint64_t t, void *stream);
/* replaces TagEnv.step (tag.py:108-143); reward is float {-1, -10, +10} */
int pomdp_tag_step(const pomdp_tag_params *p, uint32_t *state, const int32_t *action, int32_t *ob,
                   float *reward, uint8_t *done, uint32_t *err, int64_t n,
                   uint64_t seed, uint32_t lane0, uint64_t t, int flags, void *stream);

/* ---- BattleShip  (battleship.py:12-211) ------------------------------------ */
/* state: 3*MW words, MW = ceil((x_size*y_size + 6) / 32) <= 4: occupied mask words 0..MW-1, then the
 * visited mask words (cell a = y * x_size + x is bit a; total_remaining in bits 26-31 of the last
 * visited word), then the occupied mask of the lane's NEXT episode.  Board contract: whenever a board
 * is dealt at call counter t — reset() draws it from stream RESET of (lane, t); the auto-reset of a
 * step at t moves the cached next board in — the board after it is drawn from stream NEXT of (lane, t).
 * A lane's boards are therefore known one episode ahead, and a fused launch builds the boards its
 * lanes used up side by side when it ends instead of stalling a wave on one lane's rejection loop. */
typedef struct pomdp_battleship_params {
    int32_t x_size, y_size; /* x_size * y_size <= 122                     battleship.py:67 */
    int32_t max_len;        /* ctor max_len (ships max_len .. 2), 2..10   battleship.py:74-75 */
    int32_t reserved;
    /* 128-bit cell masks (4 little-endian words each) the reset's placement test uses, derived from x_size
     * and y_size on the host so that no lane recomputes them: */
    uint32_t col0[4];       /* cells with x == 0 */
    uint32_t vpat[12][4];   /* vpat[k]: k cells in a column, from cell 0 upwards (bits 0, X, 2X, ...) */
} pomdp_battleship_params;

/* replaces BattleShipEnv.reset (battleship.py:131-137, 167-211) */
int pomdp_battleship_reset(const pomdp_battleship_params *p, uint32_t *state, int32_t *ob, int64_t n,
                           uint64_t seed, uint32_t lane0, uint64_t t, void *stream);
/* replaces BattleShipEnv.step (battleship.py:91-122) */
int pomdp_battleship_step(const pomdp_battleship_params *p, uint32_t *state, const int32_t *action,
                          int32_t *ob, int32_t *reward, uint8_t *done, uint32_t *err, int64_t n,
                          uint64_t seed, uint32_t lane0, uint64_t t, int flags, void *stream);

/* ---- Tiger  (tiger.py:47-172) ---------------------------------------------- */
/* state: 1 word, bit 0 = tiger door */
typedef struct pomdp_tiger_params {
    uint64_t listen_thr;    /* listen is wrong iff k53 > listen_thr       tiger.py:141-148 */
} pomdp_tiger_params;

/* replaces TigerEnv.reset (tiger.py:60-66); the hidden state is bit 0 of the lane's word of the quad's STEP block */
int pomdp_tiger_reset(const pomdp_tiger_params *p, uint32_t *state, int32_t *ob, int64_t n,
                      uint64_t seed, uint32_t lane0, uint64_t t, void *stream);
/* replaces TigerEnv.step (tiger.py:72-88) */
int pomdp_tiger_step(const pomdp_tiger_params *p, uint32_t *state, const int32_t *action, int32_t *ob,
                     int32_t *reward, uint8_t *done, uint32_t *err, int64_t n,
                     uint64_t seed, uint32_t lane0, uint64_t t, int flags, void *stream);

/* ---- Network  (network.py:24-168) ------------------------------------------ */
/* state: 1 word, bit i = machine i is up */
typedef struct pomdp_network_params {
    int32_t  n_machines;    /* <= 32                                      network.py:27 */
    uint32_t deg_gt2_mask;  /* machines with more than 2 neighbours       network.py:89 */
    uint32_t nb_mask[32];   /* neighbour set of machine i                 network.py:144-168 */
    uint64_t fail_thr;      /* fails iff k53 >  fail_thr     (p = .1)     network.py:97 */
    uint64_t fail_nb_thr;   /* fails iff k53 >  fail_nb_thr  (q = .33)    network.py:99 */
    uint64_t obs_thr;       /* truthful iff k53 <= obs_thr   (.95)        network.py:106-109 */
} pomdp_network_params;

/* replaces NetworkEnv.reset (network.py:61-69) */
int pomdp_network_reset(const pomdp_network_params *p, uint32_t *state, int32_t *ob, int64_t n,
                        uint64_t seed, uint32_t lane0, uint64_t t, void *stream);
/* replaces NetworkEnv.step (network.py:71-114); reward = float32(float64 reward of the reference) */
int pomdp_network_step(const pomdp_network_params *p, uint32_t *state, const int32_t *action, int32_t *ob,
                       float *reward, uint8_t *done, uint32_t *err, int64_t n,
                       uint64_t seed, uint32_t lane0, uint64_t t, int flags, void *stream);

/* ---- step with bound arguments ------------------------------------------------ */
/* Everything of a pomdp_<env>_step call that does not change from one step of a batched env to the next, gathered once by
 * the caller (host memory the caller owns; the library only reads it during the call).  pomdp_step(args, action, t, stream)
 * is pomdp_<env>_step(args->params, args->state, action, args->ob, args->reward, args->done, args->err, args->n, args->seed,
 * args->lane0, t, args->flags, stream) for the env kind args->env: the same kernels, the same results — what it saves is
 * the marshalling of thirteen arguments per call in the host language's FFI (ctypes: ~1 us of a ~7 us env.step()). */
typedef struct pomdp_step_args {
    int32_t   env;       /* POMDP_ENV_* */
    int32_t   flags;     /* POMDP_AUTO_RESET or 0 */
    const void *params;  /* the env's pomdp_<env>_params (host) */
    uint32_t *state;     /* device, [words][n] */
    int32_t  *ob;        /* device [n] */
    void     *reward;    /* device [n], int32 or float per env */
    uint8_t  *done;      /* device [n] */
    uint32_t *err;       /* device uint32, may be NULL */
    int64_t   n;
    uint64_t  seed;
    uint32_t  lane0;
    uint32_t  reserved;
} pomdp_step_args;
int pomdp_step(const pomdp_step_args *args, const int32_t *action, uint64_t t, void *stream);
/* pomdp_step, then wait until its outputs can be read by the host — with pomdp_reset_sync / pomdp_stream_sync the only
 * entry points of this library that block.  For hosts that step a single env the way the reference is used (python
 * scalars in and out, batch of 1, ob / reward / done pointing at PINNED HOST memory): launch and wait cost one FFI call.
 * With n == 1 the kernel publishes its outputs through a flag in pinned host memory (system-scope release) that the host
 * polls, so the call returns as soon as the outputs are visible; the stream stays ordered and later launches need no
 * further wait.  With n != 1 it is pomdp_step followed by hipStreamSynchronize(stream). */
int pomdp_step_sync(const pomdp_step_args *args, const int32_t *action, uint64_t t, void *stream);
/* pomdp_<env>_reset (env: POMDP_ENV_*, params: its pomdp_<env>_params) followed by the same wait; `ob` in pinned host
 * memory when n == 1 */
int pomdp_reset_sync(int env, const void *params, uint32_t *state, int32_t *ob, int64_t n, uint64_t seed, uint32_t lane0,
                     uint64_t t, void *stream);
/* hipStreamSynchronize(stream) by itself */
int pomdp_stream_sync(void *stream);

/* ---- helpers ---------------------------------------------------------------- */
/* synthetic uniform random policy used by bench.py: lanes 4q..4q+3 share the Philox block
 * ctr = (q, t lo, t hi, STREAM_ACTION << 24); action = (word[lane & 3] * n_actions) >> 32.
 * lane0 must be a multiple of 4 and `action` 16-byte aligned; n is arbitrary. */
int pomdp_synthetic_actions(int32_t *action, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t,
                            uint32_t n_actions, void *stream);
/* raw generator, for known-answer tests: out (device) <- Philox4x32-10 of each of the n_blocks
 * (ctr[4], key[2]) pairs in `ctr_key` (device, uint32 [n_blocks][6]). */
int pomdp_philox_blocks(const uint32_t *ctr_key, uint32_t *out, int64_t n_blocks, void *stream);

/* Random-policy rollout driven from C: for s in [0, k_steps): pomdp_synthetic_actions at t0+s into
 * `action` (device scratch, int32[n]) with key `action_seed`, then pomdp_<env>_step at t0+s.  Exactly
 * the results a host loop over pomdp_synthetic_actions + step would produce, in k_steps + 1 launches:
 * when action_seed == seed (policy and env share the Philox key; their streams differ by stream id) the
 * policy kernel runs once, for t0, and every step launch also leaves the policy's actions for the
 * following call counter in `action` (on RockSample they ride in the cooperative reset pass); with a
 * distinct action_seed each step is a policy launch plus a step launch.  With POMDP_FUSE_STEPS in `flags` (shared key
 * only) up to pomdp_fuse_max() (256) consecutive steps run inside one launch: every step's ob / reward / done / next action is still
 * computed and written, in the same order, and the state when the launch ends, so every buffer holds what the per-step
 * launches leave, but a lane's state and action stay in registers between its steps and the launch ramp is paid once
 * per launch; the first launch derives the actions of t0 itself, so there is no policy launch at all.  Either way
 * `action` holds the actions of t0 + k_steps on return.  The caller's call counter advances by k_steps.  `params` points
 * at the env's pomdp_<env>_params; `reward` is int32 or float per env.  lane0 must be a multiple of 4 (the policy's
 * Philox block is shared by global lanes 4q .. 4q+3); n is arbitrary.  Params and pointers are checked before anything
 * is enqueued. */
int pomdp_rollout_synthetic(int env, const void *params, uint32_t *state, int32_t *action, int32_t *ob,
                            void *reward, uint8_t *done, uint32_t *err, int64_t n, uint64_t seed,
                            uint64_t action_seed, uint32_t lane0, uint64_t t0, int64_t k_steps, int flags,
                            void *stream);

/* Trajectory collection under the same policy (shared key, fused launches, POMDP_AUTO_RESET required): the k_steps
 * steps of pomdp_rollout_synthetic, but step s writes ROW s of ob / reward / done (device, [k_steps][pitch], pitch >= n
 * elements) instead of overwriting one row, and `action` (device, int32 [k_steps + 1][pitch]) receives the actions of
 * call counter t0 + s in row s (row k_steps = the actions the next call would take).  Row s of every output equals what
 * pomdp_synthetic_actions + pomdp_<env>_step at t0 + s leave in their n-element buffers; `state` ends as after the last
 * step.  This is the batched form of the reference callers' episode loops (rock.py:553-575): one launch per 256 steps,
 * each lane's state in registers, 13 bytes per lane-step written, state and first actions read once per launch.  Any
 * alignment and pitch >= n is accepted; columns on 16-byte boundaries (done: 4) with a pitch that is a multiple of 4 take the
 * launches that store 16 bytes per thread. */
int pomdp_collect_synthetic(int env, const void *params, uint32_t *state, int32_t *action, int32_t *ob, void *reward,
                            uint8_t *done, uint32_t *err, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t0,
                            int64_t k_steps, int64_t pitch, int flags, void *stream);

/* pomdp_collect_synthetic with its per-batch arguments bound once (see pomdp_step_args): pomdp_collect(a, t0, k, stream) is
 * pomdp_collect_synthetic(a->env, a->params, a->state, a->action, a->ob, a->reward, a->done, a->err, a->n, a->seed, a->lane0,
 * t0, k, a->pitch, a->flags, stream). */
typedef struct pomdp_collect_args {
    int32_t   env, flags;
    const void *params;
    uint32_t *state;
    int32_t  *action;    /* device [k + 1][pitch] */
    int32_t  *ob;        /* device [k][pitch] */
    void     *reward;
    uint8_t  *done;
    uint32_t *err;
    int64_t   n, pitch;
    uint64_t  seed;
    uint32_t  lane0, reserved;
} pomdp_collect_args;
int pomdp_collect(const pomdp_collect_args *args, uint64_t t0, int64_t k_steps, void *stream);

/* ---- trajectory layouts (ABI 11) ------------------------------------------------ */
/* pomdp_collect_synthetic writes the default layout, COLUMNS: four separate [step][pitch] arrays, the values the reference's
 * `ob, rw, done, info = env.step(action)` returns (rock.py:553-575) as int32 / float / uint8 columns.  A fused step then
 * feeds four write streams that lie whole columns apart, and how fast they drain depends on where the allocation's pages
 * happen to lie.  The two layouts below keep exactly the same information in ONE write stream:
 *
 * POMDP_LAYOUT_BLOCKED — same types, same 13 bytes per lane-step.  traj: uint8 [k_steps][pitch * 13], pitch a multiple of
 *   256 lanes; row s = pitch / 256 blocks of 3328 bytes, block q holding lanes 256 q .. 256 q + 255 of step s as
 *       action int32[256] | ob int32[256] | reward (int32 | float)[256] | done uint8[256]
 *   (action = the action TAKEN at step s; there is no row of "next actions").
 * POMDP_LAYOUT_PACKED — 4 bytes per lane-step.  traj: uint32 [k_steps][pitch]; record of (step s, lane i):
 *       action | ob << 8 | reward_code << 16 | done << 24
 *   Actions and observations of every env fit a byte (Tag's obs_cells <= 255 is checked).  reward_code: the reward itself as
 *   an int8 for RockSample / StochasticRock (-100, -10, 0, 10), Tag (-1, -10, 10), BattleShip (-10 .. cells - 1) and Tiger
 *   (-20, -1, 10: tiger.py:165-172); Network: kind * 68 + base for reward = float32(base - cost), cost = 0 / .1 / 2.5 for kind 0 (no action) /
 *   1 (ping) / 2 (reboot) (network.py:87-92, 103, 110).  pomdp_packed_reward() decodes either to the value COLUMNS holds.
 * POMDP_LAYOUT_NARROW (ABI 12) — the PACKED record's four bytes as four typed planes, 4 bytes per lane-step and nothing to
 *   decode.  traj: uint8 [k_steps][4][pitch], pitch a multiple of 4; plane 0 action uint8, 1 ob uint8, 2 reward_code (int8
 *   reward; Network: the code above, an index for a 204-entry table), 3 done uint8 — each plane of each step a typed array
 *   a consumer reads in place (plane p of step s starts (4 s + p) * pitch bytes into traj).
 * All three: POMDP_AUTO_RESET required, lane0 a multiple of 4; rows on 16-byte boundaries and n a multiple of 1024 take the
 * launches that store 16 bytes per thread, anything else the general ones; `state` ends as after the last step; the next
 * call derives its first actions from (seed, lane, t) again, so nothing else carries over. */
enum { POMDP_LAYOUT_COLUMNS = 0, POMDP_LAYOUT_BLOCKED = 1, POMDP_LAYOUT_PACKED = 2, POMDP_LAYOUT_NARROW = 3 };
int pomdp_collect_layout(int env, const void *params, uint32_t *state, void *traj, uint32_t *err, int64_t n, uint64_t seed,
                         uint32_t lane0, uint64_t t0, int64_t k_steps, int64_t pitch, int layout, int flags, void *stream);
/* the same with the per-batch arguments bound once (see pomdp_step_args) */
typedef struct pomdp_traj_args {
    int32_t   env, flags;
    int32_t   layout, reserved;
    const void *params;
    uint32_t *state;
    void     *traj;
    uint32_t *err;
    int64_t   n, pitch;
    uint64_t  seed;
    uint32_t  lane0, reserved2;
} pomdp_traj_args;
int pomdp_collect_traj(const pomdp_traj_args *args, uint64_t t0, int64_t k_steps, void *stream);
/* host-side: the reward a PACKED record's reward_code byte stands for (exactly the value the COLUMNS layout stores) */
double pomdp_packed_reward(int env, uint32_t reward_code);

/* PACKED records -> the default ABI's four columns, on the device (ABI 12): row s of `records` (uint32 [k_steps][pitch_in])
 * becomes row s of action int32 / ob int32 / reward (int32 | float per env: pomdp_packed_reward of the code) / done uint8,
 * each [k_steps][pitch_out], n lanes per row — the values pomdp_collect_synthetic writes into ob / reward / done and the
 * action TAKEN at each step (its rows 0 .. k_steps - 1).  One pass, 4 bytes read and 13 written per lane-step; rows and
 * columns on 16-byte boundaries (done: 4) with pitches that are multiples of 4 take the 16-byte accesses. */
int pomdp_decode_packed(int env, const uint32_t *records, int64_t n, int64_t k_steps, int64_t pitch_in, int32_t *action,
                        int32_t *ob, void *reward, uint8_t *done, int64_t pitch_out, void *stream);

/* ---- episode returns without a trajectory (ABI 12) ------------------------------- */
/* What the reference's callers do with the stream of `ob, rw, done, info = env.step(action)`: reduce it on the fly,
 *     r += discount * rw; discount *= .95          per step      (network.py:186-187, rock.py:569-570)
 *     eps.append(r) ... sum(eps) / len(eps)        per episode   (network.py:188-189)
 * pomdp_collect_returns runs the k_steps steps of pomdp_collect_synthetic (same policy, same draws, same final state, the
 * same launches) and keeps, per lane, only that reduction — nothing is written per step:
 *     acc  double [4][pitch]   row 0 ret       running return of the lane's current episode   (start at 0)
 *                              row 1 disc      its running discount                             (start at 1)
 *                              row 2 ret_done  return of the lane's last finished episode       (untouched until one ends)
 *                              row 3 ret_sum   sum of the returns of its finished episodes, added in the order they ended
 *     cnt  int32  [2][pitch]   row 0 episodes  number of finished episodes;  row 1 steps  number of steps taken
 * all in/out, so consecutive calls continue the same statistics.  Per step: ret += disc * reward; disc *= discount, in IEEE
 * double with separate multiply and add; a done step banks ret (ret_done = ret; ret_sum += ret; ++episodes) and the fresh
 * episode starts at ret = 0, disc = 1.  `reward` is the reference's own value: the integer rewards of RockSample / Tag /
 * BattleShip / Tiger, and for Network the float64 `base - .1` / `base - 2.5` of network.py:103, 110 (NOT the float32 the
 * reward column rounds it to).  POMDP_AUTO_RESET required; lane0 a multiple of 4; acc and cnt on 16-byte boundaries with a
 * pitch that is a multiple of 4 take the quad-per-thread launches. */
typedef struct pomdp_return_stats {
    double   discount;      /* the env's _discount (rock.py:115, tag.py:91, ...) */
    double  *acc;           /* device double [4][pitch] */
    int32_t *cnt;           /* device int32 [2][pitch] */
    int64_t  pitch;         /* >= n */
} pomdp_return_stats;
int pomdp_collect_returns(int env, const void *params, uint32_t *state, const pomdp_return_stats *stats, uint32_t *err,
                          int64_t n, uint64_t seed, uint32_t lane0, uint64_t t0, int64_t k_steps, int flags, void *stream);

/* ---- fused launches over the CALLER's actions (ABI 14) ----------------------------------------------------------------
 * The reference's callers hand step() whatever they like (rock.py:562-566, tag.py:310-312, network.py:181-187).  With the
 * actions of k_steps steps known up front — a replayed trajectory, a table policy evaluated on the device, another model's
 * output — the fused launches above run on them instead of on the synthetic policy: state in registers between steps, up to
 * pomdp_fuse_max() steps per launch, every sink.
 *     tape.actions  device uint8 [k_steps][tape.stride]: row s = the actions of call counter t0 + s, lane i at byte i
 *                   (tape.stride >= n bytes from one row to the next; the action plane of a POMDP_LAYOUT_NARROW trajectory
 *                   is such a tape with stride = 4 * pitch).  Every env's actions fit a byte.
 * A byte outside the env's action range leaves the lane untouched for that step — (ob, reward, done) = (0, 0, 0), the record
 * keeps the byte as its action, the returns sink books a step with reward 0 — and is counted in *err, as pomdp_<env>_step
 * does.  Everything else is as in the synthetic-policy entry point of the same sink: same draws (the env's streams at (seed,
 * lane, t0 + s) do not depend on where the actions come from), same rows, same final state; POMDP_AUTO_RESET required, lane0
 * a multiple of 4.  Rows on 4-byte boundaries (tape.actions and tape.stride multiples of 4) with n a multiple of 1024 and
 * 16-byte-aligned sinks take the quad-per-thread loops, anything else the general one.
 *   pomdp_collect_tape          the default columns WITHOUT `action` (the caller has it): ob int32, reward int32 | float, done
 *                               uint8, each [k_steps][pitch] — row s what pomdp_<env>_step(action = tape row s) at t0 + s returns
 *   pomdp_collect_tape_layout   POMDP_LAYOUT_BLOCKED / _PACKED / _NARROW, as pomdp_collect_layout
 *   pomdp_collect_tape_returns  the episode statistics of pomdp_collect_returns */
typedef struct pomdp_tape {
    const uint8_t *actions;
    int64_t stride;
} pomdp_tape;
int pomdp_collect_tape(int env, const void *params, uint32_t *state, const pomdp_tape *tape, int32_t *ob, void *reward,
                       uint8_t *done, uint32_t *err, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t0, int64_t k_steps,
                       int64_t pitch, int flags, void *stream);
int pomdp_collect_tape_layout(int env, const void *params, uint32_t *state, const pomdp_tape *tape, void *traj, uint32_t *err,
                              int64_t n, uint64_t seed, uint32_t lane0, uint64_t t0, int64_t k_steps, int64_t pitch, int layout,
                              int flags, void *stream);
int pomdp_collect_tape_returns(int env, const void *params, uint32_t *state, const pomdp_tape *tape,
                               const pomdp_return_stats *stats, uint32_t *err, int64_t n, uint64_t seed, uint32_t lane0,
                               uint64_t t0, int64_t k_steps, int flags, void *stream);

/* Steps per fused launch of the C-side drivers above (pomdp_rollout_synthetic with POMDP_FUSE_STEPS, pomdp_collect_*,
 * pomdp_heuristic_steps): a launch's fixed cost (kernel start, table build, drain) is paid once per this many steps.
 * pomdp_fuse_max(v) sets it for the calling process (1 <= v <= 256; v <= 0 only reads) and returns the previous value;
 * the default is POMDP_FUSE_MAX_DEFAULT.  Results never depend on it. */
#define POMDP_FUSE_MAX_DEFAULT 256
int pomdp_fuse_max(int v);
/* ... and what a trajectory collection of `env` in `layout` really runs per launch: pomdp_fuse_max(), except that the 13-byte
 * layouts (POMDP_LAYOUT_COLUMNS, _BLOCKED) of RockSample, Tag and Tiger stay at 64 — their launches are bound by the store
 * stream, and in a longer launch the waves drift further apart in the rows they write (RockSample 2.09 -> 2.55 us per step of
 * 2^20 lanes at 256 steps per launch); BattleShip and Network gain from the longer launch in every layout. */
int pomdp_fuse_steps(int env, int layout);

/* ---- planner hooks ------------------------------------- */
/* replaces <Env>._generate_legal (rock.py:273-291, tag.py:228-229, battleship.py:157-165, tiger.py:111-112,
 * network.py:130-131): list (device, int32 [n][stride], stride >= the env's action count) receives each
 * lane's legal actions in the reference's list order, padded with -1; len (device, int32 [n]) their number. */
int pomdp_legal_actions(int env, const void *params, const uint32_t *state, int32_t *list, int32_t *len,
                        int64_t n, int stride, void *stream);

/* replaces <Env>._compute_prob(action, next_state, ob) (rock.py:250-264, tag.py:209-217, battleship.py:80-89,
 * tiger.py:125-138, network.py:43-55): the likelihood of observing ob[i] after action[i] led to state column i
 * (BattleShip reads the grid, i.e. the state after the shot, as the reference does).  out: device double[n]. */
int pomdp_compute_prob(int env, const void *params, const uint32_t *state, const int32_t *action, const int32_t *ob,
                       double *out, int64_t n, void *stream);

/* Random rollouts — the simulations a POMCP-style planner runs through _set_state / _generate_legal /
 * step / _discount.  Lane i (global id lane0 + i, i < n_roots * sims_per_root) starts from
 * root_state column i / sims_per_root (uint32 [words][n_roots], read-only) and for k = 0 .. depth-1, while
 * not done:  list = _generate_legal() (all actions with POMDP_ROLLOUT_ALL_ACTIONS);
 *            a = list[(w * len(list)) >> 32], w = word k of stream ROLLOUT at (seed, lane, t0);
 *            (ob, r, done) = step(a) on stream STEP at (seed, lane, t0 + k);  ret += disc * r;  disc *= discount.
 * The return accumulates in IEEE double (separate multiply and add).  Per-lane outputs (device):
 * ret double[n], n_steps / first_action / last_ob int32[n], terminated uint8[n] (first_action = -1 for a simulation
 * that took no step; n_steps, last_ob and terminated may be NULL since ABI 14).  lane0 must be a multiple of 4
 * (RockSample's STEP blocks are shared by global lanes 4 q .. 4 q + 3 and travel within the hardware quad). */
enum { POMDP_ROLLOUT_ALL_ACTIONS = 1 };
int pomdp_rollout(int env, const void *params, const uint32_t *root_state, int64_t n_roots, int64_t sims_per_root,
                  int depth, double discount, int flags, uint64_t seed, uint32_t lane0, uint64_t t0,
                  double *ret, int32_t *n_steps, int32_t *first_action, int32_t *last_ob, uint8_t *terminated,
                  void *stream);

/* ---- the planning step built on those rollouts (ABI 14; BASELINE.json configs[4]: "a POMCP-style 1024-simulation
 * rollout per real step") ------------------------------------------------------------------------------------------
 * What the reference's hooks exist to be driven for (rock.py:243-245 _set_state, 266-291 _get_init_state /
 * _generate_legal, 115 _discount; readme.md:38-41 credits POMCP): the caller simulates from each root, turns the
 * simulations into action values and takes the best action in the real env.  Per root r, over its sims_per_root
 * simulations (simulation s of root r is lane r * sims_per_root + s of pomdp_rollout's outputs):
 *     visits[r][a] = the number of simulations whose first action was a
 *     q[r][a]      = (sum of their returns) / visits[r][a]            0.0 where visits == 0
 *     best[r]      = the action with the largest q among those with visits > 0, the lowest index on ties; -1 if none
 *     value[r]     = q[r][best[r]]                                     0.0 if best == -1
 * The reduction stays on the device — one workgroup per root, the root's returns staged through LDS, no atomics — and is
 * IEEE double with a DEFINED summation order, so that a CPU restatement reproduces it bit for bit: the root's simulations
 * are cut into chunks of POMDP_PLAN_CHUNK = 64 by simulation index; within a chunk the returns whose first action is a are
 * added in simulation-index order, starting from +0.0; the chunk sums are then added in chunk order, starting from +0.0;
 * the mean is one division.  Nothing depends on the launch geometry or on how roots are spread over GPUs: a root's
 * simulations are consecutive lanes, so whole roots never straddle a shard.
 * The real step is then pomdp_<env>_step(root_state, action = best, ...) — an ordinary step of n_roots lanes. */
#define POMDP_PLAN_CHUNK 64
typedef struct pomdp_plan_out {
    double  *q;        /* device double [n_roots][stride] */
    int32_t *visits;   /* device int32  [n_roots][stride] */
    int32_t *best;     /* device int32  [n_roots] */
    double  *value;    /* device double [n_roots]; may be NULL */
    int32_t  stride;   /* >= the env's action count (columns past it are left alone) */
    int32_t  reserved;
} pomdp_plan_out;
/* the reduction alone, over per-simulation returns / first actions the caller already has (n_actions <= 255; a
 * first_action outside [0, n_actions) — pomdp_rollout's -1 — counts for no action) */
int pomdp_plan_reduce(const double *ret, const int32_t *first_action, int64_t n_roots, int64_t sims_per_root,
                      int n_actions, const pomdp_plan_out *out, void *stream);
/* pomdp_rollout (same arguments, same lanes, same draws) followed by pomdp_plan_reduce: two launches.  sim_ret /
 * sim_first_action: device double / int32 [n_roots * sims_per_root], the caller's scratch — on return they hold the
 * simulations' returns and first actions, as pomdp_rollout writes them. */
int pomdp_plan(int env, const void *params, const uint32_t *root_state, int64_t n_roots, int64_t sims_per_root,
               int depth, double discount, int flags, uint64_t seed, uint32_t lane0, uint64_t t0,
               double *sim_ret, int32_t *sim_first_action, const pomdp_plan_out *out, void *stream);

/* ---- heuristic-policy support --------------------------- */
/* RockSample's per-rock side statistics — the Rock fields count, measured, lkv, lkw, prob_valuable of rock.py:78-86,
 * which RockEnv.step updates on every CHECK (rock.py:177-191) and `_generate_preferred` / `_select_target` read.
 * Device pointers, struct of arrays [num_rocks][n] (rock j of lane i at [j * n + i]). */
typedef struct pomdp_rock_belief {
    int32_t *count;          /* +1 per GOOD reading, -1 per BAD                     rock.py:183,187 */
    int32_t *measured;       /* number of CHECKs of this rock                       rock.py:178 */
    double  *lkv, *lkw;      /* likelihood of the readings if valuable / worthless  rock.py:184-189 */
    double  *prob_valuable;  /* .5 lkv / (.5 lkv + .5 lkw)                          rock.py:190-191 */
    uint32_t *check_ok;      /* [n] derived: bit j = measured[j] < 5 && |count[j]| < 2 && 0 < prob_valuable[j] < 1, the
                                "worth another CHECK" test of rock.py:371.  Maintained by the entry points below so
                                that the policy reads one word per lane instead of 20 bytes per rock; after writing
                                the arrays directly call pomdp_rock_belief_refresh. */
} pomdp_rock_belief;

/* fresh Rock objects (0, 0, 1., 1., .5) for every lane, or for the lanes with where[i] != 0 (device uint8, may be NULL) */
int pomdp_rock_belief_reset(const pomdp_rock_params *p, const pomdp_rock_belief *b, const uint8_t *where, int64_t n,
                            void *stream);
/* recompute check_ok from count / measured / prob_valuable (after the caller overwrote them, e.g. _set_state) */
int pomdp_rock_belief_refresh(const pomdp_rock_params *p, const pomdp_rock_belief *b, int64_t n, void *stream);
/* The part of RockEnv.step that maintains the statistics (rock.py:177-191), run after pomdp_rock_step on its outputs:
 * a lane whose action was CHECK j and whose ob != 0 updates rock j with eff(d) of the stored agent position (CHECK
 * does not move); with POMDP_AUTO_RESET a done lane gets fresh statistics (its reset() built new Rock objects),
 * without it done lanes are left alone. */
int pomdp_rock_belief_update(const pomdp_rock_params *p, const uint32_t *state, const int32_t *action, const int32_t *ob,
                             const uint8_t *done, const pomdp_rock_belief *b, int64_t n, int flags, void *stream);
/* replaces RockEnv._select_target (rock.py:389-399): the nearest (straight-line; the reference's `manhattan_distance`
 * is sqrt(dx^2+dy^2), coord.py:83-85) uncollected rock with count >= 0, lowest index on ties, -1 if none. */
int pomdp_rock_select_target(const pomdp_rock_params *p, const uint32_t *state, const pomdp_rock_belief *b,
                             int32_t *target, int64_t n, void *stream);

/* What `_generate_preferred(history)` reads from the planner's History of Transition(observation, action, reward,
 * next_observation, done) records (rock.py:525-550; tag.py:233-239 reads history.size, history[-1].action and
 * history[-1].ob), kept per lane as running sums so that no list of records has to be walked (a bounded history keeps
 * one byte per record of its window besides, see `ring`).  Device pointers:
 *   size, last_action, last_ob                                                                   int32 [n]
 *   total_sample[j] = sum over CHECK-j transitions of (+1 if next_ob GOOD, -1 if next_ob BAD)     rock.py:303-310
 *   total_move[j]   = sum over CHECK-j transitions of (+1 if next_ob GOOD, else -1 if the *previous*
 *                     observation was BAD — the reference's elif reads transition.observation)    rock.py:327-334
 * the two sums are int32 [num_rocks][n] for RockSample and unused (may be NULL, like move_ok) for the other envs. */
typedef struct pomdp_history {
    int32_t *size, *last_action, *last_ob;
    int32_t *total_sample, *total_move;
    uint32_t *move_ok;      /* [n] derived, RockSample only: bit j = total_move[j] >= 0 (the test of rock.py:335), bit 16 + j
                               = total_sample[j] > 0 (rock.py:311); maintained by pomdp_history_clear / _append /
                               pomdp_heuristic_steps — _generate_preferred reads this word, not the sums */
    /* History(max_size=k) of rock.py:533-544: append() pops the oldest record once the list holds more than k, so the
     * list settles at k + 1 records and the sums above cover that window only.  max_size = -1: unbounded (the reference's
     * default; ring and head may be NULL).  max_size >= 0: `size` stops at max_size + 1; for RockSample the window
     * itself is kept so that a record's contribution can leave the sums again: ring uint8 [max_size + 1][n], one byte per
     * kept transition (action | next_ob << 5 | (observation == BAD) << 7), head int32 [n] = the row the next transition
     * goes to (the oldest one once the ring is full).  Both may be NULL for the other envs. */
    uint8_t *ring;
    int32_t *head;
    int32_t max_size, reserved;
} pomdp_history;

/* History() — empty history for every lane / the lanes with where[i] != 0 (last_action = last_ob = -1) */
int pomdp_history_clear(int env, const void *params, const pomdp_history *h, const uint8_t *where, int64_t n,
                        void *stream);
/* history.append(Transition(observation, action, reward, next_observation, done)) per lane (rock.py:541-544).  With
 * POMDP_AUTO_RESET a done transition ends the episode and the lane's history starts over, empty. */
int pomdp_history_append(int env, const void *params, const pomdp_history *h, const int32_t *observation,
                         const int32_t *action, const int32_t *next_observation, const uint8_t *done, int64_t n,
                         int flags, void *stream);
/* replaces <Env>._generate_preferred(history) (rock.py:293-374 with use_heuristic=True, tag.py:231-243;
 * tiger.py:114-115, network.py:138-139 and BattleShip, which has none, give the legal list).  Output as
 * pomdp_legal_actions: list int32 [n][stride] in the reference's order padded with -1, len int32 [n].
 * b is read for RockSample only. */
int pomdp_preferred_actions(int env, const void *params, const uint32_t *state, const pomdp_rock_belief *b,
                            const pomdp_history *h, int32_t *list, int32_t *len, int64_t n, int stride, void *stream);
/* the caller's `np.random.choice(list)` over per-lane lists (rock.py:564): action[i] = list[i][(w * len[i]) >> 32]
 * with w the synthetic policy's word of (seed, lane0 + i, t) — the word pomdp_synthetic_actions uses; -1 if len[i] == 0 */
int pomdp_pick_actions(const int32_t *list, const int32_t *len, int stride, int32_t *action, int64_t n, uint64_t seed,
                       uint32_t lane0, uint64_t t, void *stream);

/* k_steps consecutive heuristic-policy steps, one launch each: per lane, at call counter t = t0 + s,
 *     a = choice(_generate_preferred(history))      == pomdp_preferred_actions + pomdp_pick_actions(seed, t)
 *     (ob, reward, done) = step(a)                   == pomdp_<env>_step(seed, t)
 *     side statistics, history.append(Transition(prev_ob, a, reward, ob, done))
 *                                                    == pomdp_rock_belief_update + pomdp_history_append
 *     prev_ob <- ob, or what reset() returned on a lane that auto-reset
 * with exactly the results of that five-launch sequence (the rollout loop of rock.py:557-573 for a batch).
 * prev_ob (device int32[n], in/out) starts as the observation reset() returned; action receives the chosen actions
 * (-1 on frozen lanes).  b is read for RockSample only.  The caller's call counter advances by k_steps.
 * `returns` (may be NULL) adds the loop's discounted return, `r += rw * discount; discount *= env._discount`
 * (rock.py:569-570), in IEEE double with separate multiply and add: ret[i] += disc[i] * reward; disc[i] *= discount.
 * When a lane's episode ends, ret_done[i] receives its return; with POMDP_AUTO_RESET ret[i] / disc[i] then restart at
 * 0 / 1 for the new episode, without it they keep the finished episode's values (the lane is frozen).
 * lane0 must be a multiple of 4 (the policy's block is shared by global lanes 4 q .. 4 q + 3). */
typedef struct pomdp_returns {
    double  discount;       /* the env's _discount (rock.py:115, tag.py:91, ...) */
    double *ret, *disc;     /* device double[n], in/out; start them at 0 and 1 */
    double *ret_done;       /* device double[n], out: return of the last finished episode of the lane */
} pomdp_returns;
int pomdp_heuristic_steps(int env, const void *params, uint32_t *state, const pomdp_rock_belief *b, const pomdp_history *h,
                          int32_t *prev_ob, int32_t *action, int32_t *ob, void *reward, uint8_t *done,
                          const pomdp_returns *returns, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t0,
                          int64_t k_steps, int flags, void *stream);

int         pomdp_abi_version(void);
const char *pomdp_error_string(int code);
/* introspection: the kernel (name<template arguments>, as a profiler shows it) that the calling thread's most recent
 * pomdp_rollout_synthetic(POMDP_FUSE_STEPS) / pomdp_collect_synthetic launch picked for the batch geometry it was given;
 * "" before the first such call.  The string is thread-local and overwritten by the next call. */
const char *pomdp_last_fused_kernel(void);

#ifdef __cplusplus
}
#endif
#endif
