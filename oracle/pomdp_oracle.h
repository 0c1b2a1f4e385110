/*
 * pomdp_oracle.h — CPU restatement of d3sm0/gym_pomdp's reset()/step() semantics.
 *
 * TEST INFRASTRUCTURE.  This library is the parity oracle for the HIP path in
 * gym_pomdp_amd/csrc.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product path never does and fails loudly
 * when its HIP extension is missing.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py)
 * against golden traces generated in the build container by importing the
 * unmodified reference (tests/golden/generate_golden.py):
 *   mode A  the reference on its native np.random.seed(s) MT19937 stream,
 *   mode B  the reference with its MT19937 state overwritten so that numpy
 *           derives its draws from this build's Philox4x32-10 word stream.
 * Every function below cites the reference file:line it follows.
 */
#ifndef POMDP_ORACLE_H
#define POMDP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- 32-bit word sources ------------------------------------------------ */
enum { OR_WS_MT19937 = 0, OR_WS_PHILOX = 1 };
enum { OR_STREAM_STEP = 0, OR_STREAM_RESET = 1, OR_STREAM_STEP_SPACE = 2,
       OR_STREAM_RESET_SPACE = 3, OR_STREAM_ACTION = 4, OR_STREAM_ROLLOUT = 5, OR_STREAM_NEXT = 6, OR_STREAM_STEP_LO = 7 };

typedef struct or_ws {
    int kind;
    /* MT19937 (numpy legacy RandomState, np.random.seed(int)) */
    uint32_t mt[624];
    int mti;
    /* Philox4x32-10 stream: key=(seed lo,hi) ctr=(lane,t lo,t hi,stream<<24|blk) */
    uint32_t key[2];
    uint32_t ctr[4];
    uint32_t blk[4];
    uint32_t widx;
    uint64_t n_drawn;
    /* RockSample's split layout (oracle/philox_ref.py: rock_reset_words / rock_step_words):
     * 0 = plain sequential stream, 1 = per-lane split, 2 = quad-shared split (STEP), 3 = rotated pair (RESET), 4 = Network's
     * STEP: top 16 bits of double j from the quad's block j >> 1, the rest from the lane's STEP_LO block j >> 1; 5 / 6 = Tag with
     * one opponent: a step's flight / the auto-reset after it, from the quad's STEP word (pomdp_oracle.c: or_ws_next32) */
    int layout;
    uint32_t blk_base;                      /* layout 3: first block of the rotated pair (auto-reset: the step's sensor block) */
    uint32_t lane;
    uint32_t half_blk[2][4], half_idx[2];   /* one cached block per half (high / low words) */
    int half_have[2];
} or_ws;

void     or_ws_seed_mt(or_ws *ws, uint32_t seed);
void     or_ws_philox(or_ws *ws, uint64_t seed, uint32_t lane, uint64_t t, uint32_t stream);
/* word source of one env call: RockSample envs get the split layout for streams STEP / RESET */
void     or_ws_philox_env(or_ws *ws, int env_kind, uint64_t seed, uint32_t lane, uint64_t t, uint32_t stream);
/* Tiger's gym-space RNG at call counter t: the quad's STEP blocks (oracle/philox_ref.py tiger_words) */
void     or_ws_space(or_ws *ws, uint64_t seed, uint32_t lane, uint64_t t, uint32_t stream);
uint32_t or_ws_next32(or_ws *ws);
void     or_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* numpy legacy constructions on top of a word source */
uint64_t or_draw_k53(or_ws *ws);           /* U = k / 2^53, k returned      */
uint32_t or_draw_randint(or_ws *ws, uint32_t n); /* np.random.randint(n)    */

/* ---- envs ---------------------------------------------------------------- */
enum { OR_ENV_ROCK = 0, OR_ENV_TAG = 1, OR_ENV_BATTLESHIP = 2, OR_ENV_TIGER = 3, OR_ENV_NETWORK = 4 };
enum { OR_REWARD_I32 = 0, OR_REWARD_F32 = 1 };

typedef struct or_env or_env;
/* word source of the reset that follows a done step inside that step's call counter t (batch auto-reset) */
void     or_ws_philox_auto_reset(or_ws *ws, const or_env *e, uint64_t seed, uint32_t lane, uint64_t t);
/* np.random's word source inside step() at call counter t (Tag with one opponent: the quad's word, layout 5) */
void     or_ws_philox_step(or_ws *ws, const or_env *e, uint64_t seed, uint32_t lane, uint64_t t);

/* args: rock (board_size, num_rocks[, stochastic, act_thr_lo, act_thr_hi])  stochastic = StochasticRockEnv
 *       stochrock (board_size, num_rocks, 1, act_thr_lo, act_thr_hi[, act_gt])  thr==0 -> the captured value for 0.8; act_gt: acts iff k > thr
 *       tag (num_opponents, obs_cells, move_thr_lo, move_thr_hi[, move_gt])  thr==0 -> captured value for 0.8; move_gt: moves iff k > thr
 *       battleship (x_size, y_size, max_len)
 *       tiger ()
 *       network (n_machines, problem_type)
 * Returns NULL for configurations the reference rejects at construction. */
or_env *or_env_new(int kind, const int64_t *args, int nargs);
or_env *or_env_clone(const or_env *e);
void    or_env_free(or_env *e);
int     or_env_n_actions(const or_env *e);
int     or_env_n_obs(const or_env *e);
int     or_env_compact_len(const or_env *e);
int     or_env_words(const or_env *e);        /* packed int32 words per lane (GPU layout) */
int     or_env_reward_kind(const or_env *e);

int  or_env_reset(or_env *e, or_ws *np_rng, or_ws *space_rng);   /* returns ob */
void or_env_step(or_env *e, int action, or_ws *np_rng, or_ws *space_rng,
                 int *ob, double *reward, int *done);
void or_env_compact(const or_env *e, int64_t *out);
void or_env_pack(const or_env *e, uint32_t *words);
void or_env_unpack(or_env *e, const uint32_t *words);

/* ---- mode A: sequential single-env trace on MT19937 ----------------------- */
/* out_* arrays have T entries (state arrays T*compact_len).  reset() is called
 * right after each done step.  Returns 0, or -1 on invalid action. */
int or_trace_mt(or_env *e, uint32_t seed, uint32_t space_seed, const int64_t *actions, int64_t T,
                int64_t *ob0, int64_t *state0, int64_t *ob, double *reward, uint8_t *done,
                int64_t *state_pre, int64_t *state, int64_t *reset_ob);

/* ---- mode B: batched lanes on Philox streams (GPU-parity + CPU baseline) --- */
/* state: uint32 [words][n] SoA.  reward: int32[n] or float[n] per reward kind. */
void or_batch_reset(const or_env *proto, uint32_t *state, int32_t *ob, int64_t n,
                    uint64_t seed, uint32_t lane0, uint64_t t, int nthreads);
/* returns the number of lanes whose action was out of range (treated as no-op) */
int64_t or_batch_step(const or_env *proto, uint32_t *state, const int32_t *action, int32_t *ob,
                      void *reward, uint8_t *done, int64_t n, uint64_t seed, uint32_t lane0,
                      uint64_t t, int auto_reset, int nthreads);
/* compact (reference-format) state of every lane: out[n][compact_len] */
void or_batch_compact(const or_env *proto, const uint32_t *state, int64_t *out, int64_t n);
void or_synthetic_actions(int32_t *action, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t,
                          uint32_t n_actions, int nthreads);
int  or_max_threads(void);
/* CPU baseline loop, entirely in C: reset n lanes, then `steps` x (synthetic actions + step with auto-reset)
 * on `nthreads` OpenMP threads; returns the wall-clock seconds of the stepping part and the number of done
 * lanes seen (so the work cannot be optimised away). */
double or_bench_loop(const or_env *proto, int64_t n, int64_t steps, uint64_t seed, int nthreads, int64_t *n_done);

/* The reference callers' reduction of a random-policy rollout (network.py:175-191, rock.py:553-575): per lane and step
 * ret += disc * reward; disc *= discount (IEEE double, separate multiply and add); a done step banks ret — ret_done = ret,
 * ret_sum += ret, episodes++ — and the fresh episode starts at 0 / 1.  k steps of or_synthetic_actions + or_batch_step
 * (auto_reset) from call counter t0; `actions` (int32 [k][n], may be NULL) replaces the policy's.  acc: double [4][pitch] (ret, disc, ret_done, ret_sum), cnt: int32 [2][pitch]
 * (episodes, steps), in/out.  The reward is the reference's float64 value.  Returns the number of done steps. */
int64_t or_batch_collect_returns(const or_env *proto, uint32_t *state, double *acc, int32_t *cnt, int64_t pitch, double discount,
                                 const int32_t *actions, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t0, int64_t k,
                                 int nthreads);

/* ---- planner hooks ---------------------------------- */
/* `_generate_legal()` of the current state, in the reference's list order (duplicates kept);
 * returns the list length (<= OR_MAX_LEGAL). */
#define OR_MAX_LEGAL 160
int  or_env_legal(const or_env *e, int *list);
/* legal lists of every lane of a packed batch: out[n][OR_MAX_LEGAL] padded with -1, len[n] */
void or_batch_legal(const or_env *proto, const uint32_t *state, int32_t *out, int32_t *len, int64_t n);
/* `_compute_prob(action, next_state, ob)` with next_state = the env's current state
 * (rock.py:250-264, tag.py:209-217, battleship.py:80-89, tiger.py:125-138, network.py:43-55) */
double or_env_compute_prob(const or_env *e, int action, int ob);
void   or_batch_compute_prob(const or_env *proto, const uint32_t *state, const int32_t *action, const int32_t *ob,
                             double *out, int64_t n);
/* Random rollouts (POMCP-style simulations).  Lane i starts from root state column i / sims_per_root
 * (state: uint32 [words][n_roots], read-only) and, for k = 0 .. depth-1 while not done:
 *   list = policy ? all actions : _generate_legal();  w = word k of stream ROLLOUT at (seed, lane, t0)
 *   a = list[(w * len(list)) >> 32];  (ob, r, done) = step(a) on stream STEP at (seed, lane, t0+k)
 *   ret += disc * r;  disc *= discount      (IEEE double, separate multiply and add)
 * Outputs per lane: ret (double), n_steps, first action, last observation, terminated flag. */
void or_batch_rollout(const or_env *proto, const uint32_t *state, int64_t n_roots, int64_t sims_per_root,
                      int depth, double discount, int policy_all_actions, uint64_t seed, uint32_t lane0,
                      uint64_t t0, double *ret, int32_t *n_steps, int32_t *first_action, int32_t *last_ob,
                      uint8_t *terminated, int nthreads);

/* The planning step a POMCP-style caller builds on those rollouts (BASELINE.json configs[4]; the hooks it drives:
 * rock.py:243-245 _set_state, 266-291 _get_init_state / _generate_legal, 115 _discount): per root,
 * over its sims_per_root simulations (simulation s of root r = lane r * sims_per_root + s of or_batch_rollout's outputs),
 *   visits[r][a] = the number of simulations whose first action was a
 *   q[r][a]      = (sum of their returns) / visits[r][a]           0.0 where visits == 0
 *   best[r]      = the action with the largest q among those with visits > 0, lowest index on ties; -1 if there is none
 *   value[r]     = q[r][best[r]]                                    0.0 if best == -1
 * in IEEE double with this summation order, which the GPU build states in include/pomdp_hip.h (pomdp_plan) and follows:
 * the root's simulations are cut into chunks of 64 by simulation index; a chunk's returns for action a are added in
 * simulation-index order starting from +0.0; the chunk sums are then added in chunk order starting from +0.0.
 * q / visits: [n_roots][stride], stride >= n_actions (columns past n_actions are left alone). */
#define OR_PLAN_CHUNK 64
void or_plan_reduce(const double *ret, const int32_t *first_action, int64_t n_roots, int64_t sims_per_root, int n_actions,
                    int stride, double *q, int32_t *visits, int32_t *best, double *value);


/* ---- heuristic-policy support ------------------------ */
/* RockSample's per-rock side statistics (rock.py:78-86: count, measured, lkw, lkv, prob_valuable), struct of
 * arrays [num_rocks][n].  The reference updates them inside step() on every CHECK (rock.py:177-191). */
typedef struct or_rock_belief {
    int32_t *count, *measured;
    double *lkv, *lkw, *prob_valuable;
} or_rock_belief;
/* where == NULL: every lane; otherwise lanes with where[i] != 0.  Fresh Rock objects: 0, 0, 1., 1., .5 */
void or_batch_rock_belief_reset(const or_env *proto, const or_rock_belief *b, const uint8_t *where, int64_t n);
/* after a step: `state` is the stored (post auto-reset) state, (action, ob, done) what the step returned.
 * A lane that executed CHECK j (ob != 0) updates rock j as rock.py:177-191 does; with auto_reset a done lane's
 * statistics are those of a fresh episode. */
void or_batch_rock_belief_update(const or_env *proto, const uint32_t *state, const int32_t *action, const int32_t *ob,
                                 const uint8_t *done, int auto_reset, const or_rock_belief *b, int64_t n);

/* What `_generate_preferred(history)` reads from the planner's History (rock.py:525-550; tag.py:233-239 reads
 * history.size and history[-1].action / .ob), kept as running sums so the history itself need not be stored:
 *   size, last_action = history[-1].action, last_ob = history[-1].next_observation             [n]
 *   total_sample[j] = sum over transitions with action == CHECK j of (+1 next_ob GOOD, -1 next_ob BAD)   rock.py:303-310
 *   total_move[j]   = same transitions: +1 if next_ob GOOD, else -1 if *observation* is BAD              rock.py:327-334
 * (the two rock sums, [num_rocks][n], are NULL for the other envs).
 * Bounded histories (History(max_size=k), rock.py:533-544) are kept the way the reference keeps them — a list of the
 * records themselves, oldest first, from which append() pops element 0 when size > max_size BEFORE appending (so the
 * list settles at k + 1 records) — in rec_obs / rec_act / rec_next [max_size + 1][n]; `size` is then the list length and
 * the two sums are taken over the stored records each time they are asked for.  max_size < 0: unbounded. */
typedef struct or_history {
    int32_t *size, *last_action, *last_ob;
    int32_t *total_sample, *total_move;
    int32_t max_size, reserved;
    int32_t *rec_obs, *rec_act, *rec_next;
} or_history;
void or_batch_history_clear(const or_env *proto, const or_history *h, const uint8_t *where, int64_t n);
/* history.append(Transition(observation, action, reward, next_observation, done)); with auto_reset a done
 * transition ends the episode and the lane's history starts over (empty). */
void or_batch_history_append(const or_env *proto, const or_history *h, const int32_t *observation,
                             const int32_t *action, const int32_t *next_observation, const uint8_t *done,
                             int auto_reset, int64_t n);
/* `_generate_preferred(history)` with use_heuristic=True (rock.py:293-374, tag.py:231-243; tiger.py:114-115 and
 * network.py:138-139 return the legal list; BattleShip has no such method and gets its legal list):
 * out[n][OR_MAX_LEGAL] padded with -1, len[n].  b may be NULL for non-rock envs. */
void or_batch_preferred(const or_env *proto, const uint32_t *state, const or_rock_belief *b, const or_history *h,
                        int32_t *out, int32_t *len, int64_t n);
/* RockEnv._select_target (rock.py:389-399): nearest (straight-line distance — the reference's
 * `manhattan_distance` is sqrt(dx^2+dy^2), coord.py:83-85) uncollected rock with count >= 0, first wins; -1 if none */
void or_batch_rock_select_target(const or_env *proto, const uint32_t *state, const or_rock_belief *b,
                                 int32_t *target, int64_t n);
/* action[i] = list[i][(w * len[i]) >> 32], w = the synthetic policy's word of (seed, lane0 + i, t) (stream ACTION,
 * shared by the four lanes of a quad like or_synthetic_actions); -1 where len[i] == 0 */
void or_batch_pick(const int32_t *list, const int32_t *len, int stride, int32_t *action, int64_t n, uint64_t seed,
                   uint32_t lane0, uint64_t t);

/* k steps of the reference's heuristic rollout loop (rock.py:557-573) for every lane in one call, lane-major, built from
 * the per-lane pieces of the batch functions above; rows [k][n] out, state / b / h / prev_ob updated in place */
void or_batch_heuristic_steps(const or_env *proto, uint32_t *state, const or_rock_belief *b, const or_history *h,
                              int32_t *prev_ob, const uint8_t *done_in, int32_t *action, int32_t *ob, void *reward,
                              uint8_t *done, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t0, int64_t k,
                              int auto_reset, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
