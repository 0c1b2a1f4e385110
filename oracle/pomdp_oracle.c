/*
 * pomdp_oracle.c — CPU restatement of d3sm0/gym_pomdp's reset()/step().
 * TEST INFRASTRUCTURE (see pomdp_oracle.h).  Plain C, scalar, one env object
 * per lane, written to read like the reference's Python; every block cites the
 * reference file:line it follows (paths relative to gym_pomdp/envs/).
 */
#include "pomdp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ======================================================================== */
/* word sources                                                              */
/* ======================================================================== */

/* MT19937 (Matsumoto & Nishimura 1998) as numpy's legacy RandomState uses it:
 * np.random.seed(int s) == init_genrand(s)  [SURVEY.md §8c, verified]. */
void or_ws_seed_mt(or_ws *ws, uint32_t seed)
{
    memset(ws, 0, sizeof(*ws));
    ws->kind = OR_WS_MT19937;
    ws->mt[0] = seed;
    for (int i = 1; i < 624; i++)
        ws->mt[i] = 1812433253u * (ws->mt[i - 1] ^ (ws->mt[i - 1] >> 30)) + (uint32_t)i;
    ws->mti = 624;
}

static void mt_refill(or_ws *ws)
{
    uint32_t *mt = ws->mt;
    for (int k = 0; k < 624; k++) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        uint32_t v = mt[(k + 397) % 624] ^ (y >> 1);
        if (y & 1u) v ^= 0x9908b0dfu;
        mt[k] = v;
    }
    ws->mti = 0;
}

void or_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void or_ws_philox(or_ws *ws, uint64_t seed, uint32_t lane, uint64_t t, uint32_t stream)
{
    ws->kind = OR_WS_PHILOX;
    ws->key[0] = (uint32_t)seed;
    ws->key[1] = (uint32_t)(seed >> 32);
    ws->ctr[0] = lane;
    ws->ctr[1] = (uint32_t)t;
    ws->ctr[2] = (uint32_t)(t >> 32);
    ws->ctr[3] = stream << 24;
    ws->widx = 0;
    ws->n_drawn = 0;
    ws->layout = 0;
    ws->blk_base = 0;
    ws->lane = lane;
    ws->half_have[0] = ws->half_have[1] = 0;
}

/* RockSample: every draw is a double, double j = (high word, low word) with the two halves in different Philox
 * blocks (split layout); both streams are shared by the four lanes of a quad (counter word 0 = lane >> 2).  See
 * oracle/philox_ref.py rock_reset_words / rock_step_words for the normative statement. */
void or_ws_philox_env(or_ws *ws, int env_kind, uint64_t seed, uint32_t lane, uint64_t t, uint32_t stream)
{
    or_ws_philox(ws, seed, lane, t, stream);
    if (env_kind == OR_ENV_ROCK && stream == OR_STREAM_RESET) { ws->layout = 3; ws->ctr[0] = lane >> 2; }   /* quad-shared, rotated */
    if (env_kind == OR_ENV_ROCK && stream == OR_STREAM_STEP) { ws->layout = 2; ws->ctr[0] = lane >> 2; }
    /* Network: every draw of step() is a double too (one per up machine, one for the action): the top 16 bits of double j
     * from a quad-shared block, the rest from the lane's own STEP_LO stream (oracle/philox_ref.py network_step_words) */
    if (env_kind == OR_ENV_NETWORK && stream == OR_STREAM_STEP) ws->layout = 4;
    if (env_kind == OR_ENV_TIGER && stream == OR_STREAM_STEP) { ws->layout = 2; ws->ctr[0] = lane >> 2; }
}

/* Tiger's gym-space RNG (state_space.sample(): tiger.py:62, 118) — ABI 13: it reads what LISTEN's uniform() reads, the
 * quad's STEP blocks of the call counter (oracle/philox_ref.py tiger_words); `stream` (STEP_SPACE / RESET_SPACE) only says
 * which call site draws */
void or_ws_space(or_ws *ws, uint64_t seed, uint32_t lane, uint64_t t, uint32_t stream)
{
    (void)stream;
    or_ws_philox_env(ws, OR_ENV_TIGER, seed, lane, t, OR_STREAM_STEP);
}

uint32_t or_ws_next32(or_ws *ws)
{
    ws->n_drawn++;
    if (ws->kind == OR_WS_MT19937) {
        if (ws->mti >= 624) mt_refill(ws);
        uint32_t y = ws->mt[ws->mti++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    if (ws->layout == 5 || ws->layout == 6) {
        /* Tag with one opponent (oracle/philox_ref.py tag_step_words / tag_auto_reset_words): W = element lane & 3 of block 0 of
         * the QUAD's STEP stream, W' the same of block 1.  layout 5 (a step: the flight's binomial double, then the choice's
         * word): W, W', W.  layout 6 (the auto-reset after a done step): attempt i < 6 reads W >> 5 i, later attempts the
         * lane's own RESET stream from its first word on. */
        const uint32_t i = ws->widx++;
        if (ws->layout == 6 && i >= 6u) {
            const uint32_t k = i - 6u;
            if ((k & 3u) == 0 || !ws->half_have[1] || ws->half_idx[1] != (k >> 2)) {
                uint32_t c[4] = { ws->lane, ws->ctr[1], ws->ctr[2], ((uint32_t)OR_STREAM_RESET << 24) | ((k >> 2) & 0xFFFFFFu) };
                or_philox4x32_10(c, ws->key, ws->half_blk[1]);
                ws->half_idx[1] = k >> 2;
                ws->half_have[1] = 1;
            }
            return ws->half_blk[1][k & 3u];
        }
        const uint32_t block = (ws->layout == 5 && i == 1u) ? 1u : 0u;
        if (ws->layout == 5 && i > 2u) return 0xDEADBEEFu;                   /* a flight draws three words */
        uint32_t c[4] = { ws->lane >> 2, ws->ctr[1], ws->ctr[2], ((uint32_t)OR_STREAM_STEP << 24) | block }, q[4];
        or_philox4x32_10(c, ws->key, q);
        const uint32_t w = q[ws->lane & 3u];
        return ws->layout == 6 ? (w >> (5u * i)) : w;
    }
    if (ws->layout == 4) {
        /* word 2 j = Q_j << 16 | X_j >> 16, word 2 j + 1 = Y_j: Q_j = the upper (j even) / lower (j odd) half of element
         * lane & 3 of block j >> 1 of the QUAD's STEP stream; X_j, Y_j = elements 2 (j & 1), 2 (j & 1) + 1 of block j >> 1
         * of the lane's STEP_LO stream.  half_blk[0] caches the quad block, half_blk[1] the lane's. */
        const uint32_t i = ws->widx++, j = i >> 1, second = i & 1u, block = j >> 1;
        if (!ws->half_have[0] || ws->half_idx[0] != block) {
            uint32_t c[4] = { ws->lane >> 2, ws->ctr[1], ws->ctr[2], ((uint32_t)OR_STREAM_STEP << 24) | block };
            or_philox4x32_10(c, ws->key, ws->half_blk[0]);
            uint32_t c2[4] = { ws->lane, ws->ctr[1], ws->ctr[2], ((uint32_t)OR_STREAM_STEP_LO << 24) | block };
            or_philox4x32_10(c2, ws->key, ws->half_blk[1]);
            ws->half_idx[0] = block;
            ws->half_have[0] = 1;
        }
        const uint32_t w = ws->half_blk[0][ws->lane & 3u], q = (j & 1u) ? (w & 0xFFFFu) : (w >> 16);
        const uint32_t x = ws->half_blk[1][2u * (j & 1u)], y = ws->half_blk[1][2u * (j & 1u) + 1u];
        return second ? y : ((q << 16) | (x >> 16));
    }
    if (ws->layout != 0) {
        const uint32_t i = ws->widx++, j = i >> 1, half = i & 1u;        /* word i = half `half` of double j */
        /* layout 1: per-lane split (four doubles per block pair); 2: quad-shared split (one double per lane per pair);
         * 3: RockSample reset — quad-shared like 2, ONE block pair: double j (rock j) = the lane's element of block
         *    `half` rotated right by 2 j + 2 bits (its top bit is bit 2 j + 1 of the element) */
        const uint32_t block = ws->layout == 1 ? 2u * (j >> 2) + half : ws->layout == 3 ? ws->blk_base + half : 2u * j + half;
        const uint32_t elem = ws->layout == 1 ? (j & 3u) : (ws->lane & 3u);
        const uint32_t rot = ws->layout == 3 ? ((2u * j + 2u) & 31u) : 0u;
        if (!ws->half_have[half] || ws->half_idx[half] != block) {
            uint32_t c[4] = { ws->ctr[0], ws->ctr[1], ws->ctr[2], ws->ctr[3] | block };
            or_philox4x32_10(c, ws->key, ws->half_blk[half]);
            ws->half_idx[half] = block;
            ws->half_have[half] = 1;
        }
        const uint32_t w = ws->half_blk[half][elem];
        return rot ? ((w >> rot) | (w << (32u - rot))) : w;
    }
    if ((ws->widx & 3u) == 0) {
        uint32_t c[4] = { ws->ctr[0], ws->ctr[1], ws->ctr[2], ws->ctr[3] | ((ws->widx >> 2) & 0xFFFFFFu) };
        or_philox4x32_10(c, ws->key, ws->blk);
    }
    return ws->blk[ws->widx++ & 3u];
}

/* numpy legacy next_double: (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53 */
uint64_t or_draw_k53(or_ws *ws)
{
    uint64_t a = or_ws_next32(ws) >> 5;
    uint64_t b = or_ws_next32(ws) >> 6;
    return (a << 26) + b;
}

/* np.random.randint(n) on the legacy stream: smallest all-ones mask >= n-1,
 * one 32-bit word per attempt, reject while (word & mask) > n-1; n == 1 draws nothing. */
uint32_t or_draw_randint(or_ws *ws, uint32_t n)
{
    uint32_t rng = n - 1;
    if (rng == 0) return 0;
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    do { v = or_ws_next32(ws) & mask; } while (v > rng);
    return v;
}

/* ======================================================================== */
/* captured Bernoulli thresholds (tests/golden/thresholds.json, fixture F2)  */
/* np.random.binomial(1, p) consumes one double U = k/2^53 and returns       */
/*   p >  .5 : 1 iff k <= thr          p <= .5 : 1 iff k > thr               */
/* ======================================================================== */
#define TWO52 4503599627370496ULL
static const uint64_t ROCK_THR[29] = {
    9007199254740992ULL, 8853790118380056ULL, 8705606660380041ULL, 8562470874952118ULL,
    8424210819838096ULL, 8290660409764310ULL, 8161659216931231ULL, 8037052278299120ULL,
    7916689909438254ULL, 7800427524720092ULL, 7688125463633382ULL, 7579648823016592ULL,
    7474867295005110ULL, 7373655010498564ULL, 7275890387960212ULL, 7181455987366794ULL,
    7090238369133370ULL, 7002127957843708ULL, 6917018910622514ULL, 6834808989991382ULL,
    6755399441055744ULL, 6678694872875276ULL, 6604603143875268ULL, 6533035251161307ULL,
    6463905223604296ULL, 6397130018567403ULL, 6332629422150863ULL, 6270325952834808ULL,
    6210144768404375ULL };
#define TAG_MOVE_THR   7205759403792794ULL /* binomial(1, .8)  : 1 iff k <= thr */
#define NET_FAIL_THR   8106479329266893ULL /* binomial(1, .1)  : 1 iff k >  thr */
#define NET_FAILNB_THR 6034823500676464ULL /* binomial(1, .33) : 1 iff k >  thr */
#define NET_OBS_THR    8556839292003942ULL /* binomial(1, .95) : 1 iff k <= thr */
#define TIGER_THR      7656119366529843ULL /* uniform() > .85  : iff k > thr    */

/* ======================================================================== */
/* env objects                                                               */
/* ======================================================================== */
#define MAX_ROCKS 16
#define MAX_OPP 4
#define MAX_CELLS 128
#define MAX_MACH 32

typedef struct { int x, y; } coord;

/* coord.py:101-110  Moves: NORTH (0,1) EAST (1,0) SOUTH (0,-1) WEST (-1,0) */
static const coord MOVES[4] = { {0, 1}, {1, 0}, {0, -1}, {-1, 0} };
/* battleship.py:12-21  Compass, enumeration order */
static const coord COMPASS[9] = { {0, 1}, {1, 0}, {0, -1}, {-1, 0}, {0, 0}, {1, 1}, {1, -1}, {-1, -1}, {-1, 1} };

struct or_env {
    int kind;
    /* ---- rock ---- */
    int size, num_rocks, n_listed;
    int stochastic;            /* StochasticRockEnv (rock.py:428-504): action applied w.p. p_move, penalty 0 */
    uint64_t act_thr;          /* binomial(1, p_move) == 1 iff k53 <= act_thr */
    int act_gt;                /* ... iff k53 > act_thr for p_move <= .5 (numpy's inversion takes the other branch) */
    coord start, rock_pos[MAX_ROCKS];
    int grid[16][16];          /* grid.board[x, y], -1 = empty (rock.py:108-111) */
    coord agent;
    int status[MAX_ROCKS];     /* Rock.status in {-1, 0, +1} */
    /* ---- tag ---- */
    int n_opponents, obs_cells;
    uint64_t move_thr;
    int move_gt;            /* numpy's binomial(1, p) is [U > thr] for p <= .5, [U <= thr] for p > .5 */
    coord opp[MAX_OPP];
    int num_opp;
    /* ---- battleship ---- */
    int xs, ys, max_len;       /* max_len = ctor max_len + 1 (battleship.py:75) */
    uint8_t occ[16][16], vis[16][16];
    uint8_t next_occ[16][16];  /* batched board contract: the NEXT episode's ships (see bs_deal_next) */
    int remaining;
    /* ---- tiger ---- */
    int tiger;
    /* ---- network ---- */
    int n_mach, nb_len[MAX_MACH], nb[MAX_MACH][4];
    int up[MAX_MACH];
};

/* rock.py:43-64 */
typedef struct { int size, k_a, k_b, sx, sy, n; int pos[16][2]; } rock_cfg;
static const rock_cfg ROCK_CFGS[5] = {
    { 2, 2, 1, 0, 0, 1, { {1, 0} } },
    { 4, 4, 3, 0, 0, 3, { {1, 0}, {3, 1}, {2, 3} } },
    { 7, 7, 8, 0, 3, 8, { {2, 0}, {0, 1}, {3, 1}, {6, 3}, {2, 4}, {3, 4}, {5, 5}, {1, 6} } },
    { 11, 11, 11, 0, 5, 11,
      { {0, 3}, {0, 7}, {1, 8}, {2, 4}, {3, 3}, {3, 8}, {4, 3}, {5, 8}, {6, 1}, {9, 3}, {9, 9} } },
    { 15, 15, 15, 0, 5, 16,
      { {0, 7}, {0, 3}, {1, 2}, {1, 2}, {2, 6}, {3, 7}, {3, 2}, {4, 7}, {5, 2}, {6, 9}, {9, 7}, {9, 1},
        {11, 8}, {13, 10}, {14, 9}, {12, 2} } },
};

static int rock_init(or_env *e, int board_size, int num_rocks)
{
    const rock_cfg *c = NULL;
    for (int i = 0; i < 5; i++) if (ROCK_CFGS[i].size == board_size) c = &ROCK_CFGS[i];
    /* rock.py:101  assert board_size in config and num_rocks in config[bs]['size'] */
    if (!c || (num_rocks != c->k_a && num_rocks != c->k_b)) return -1;
    /* (2,2) and (4,4) pass the assert but reset() raises IndexError: rejected here */
    if (num_rocks > c->n) return -1;
    e->size = board_size; e->num_rocks = num_rocks; e->n_listed = c->n;
    e->start.x = c->sx; e->start.y = c->sy;
    for (int x = 0; x < 16; x++) for (int y = 0; y < 16; y++) e->grid[x][y] = -1;
    for (int i = 0; i < c->n; i++) {            /* rock.py:110-111: every listed coord is stamped */
        e->rock_pos[i].x = c->pos[i][0]; e->rock_pos[i].y = c->pos[i][1];
        e->grid[c->pos[i][0]][c->pos[i][1]] = i;
    }
    return 0;
}

/* network.py:144-168 */
static int network_init(or_env *e, int n, int problem_type)
{
    if (n < 1 || n > MAX_MACH) return -1;
    e->n_mach = n;
    memset(e->nb_len, 0, sizeof(e->nb_len));
    if (problem_type == 3) {                     /* make_3legs_neighbours */
        if (!(n >= 4 && n % 3 == 1)) return -1;
        e->nb[0][0] = 1; e->nb[0][1] = 2; e->nb[0][2] = 3; e->nb_len[0] = 3;
        for (int i = 1; i < n; i++) {
            if (i < n - 3) e->nb[i][e->nb_len[i]++] = i + 3;
            if (i <= 4) e->nb[i][e->nb_len[i]++] = 0;
            else e->nb[i][e->nb_len[i]++] = i - 3;
        }
    } else {                                     /* make_ring_neighbours */
        for (int i = 0; i < n; i++) {
            e->nb[i][0] = (i + 1) % n; e->nb[i][1] = (i + n - 1) % n; e->nb_len[i] = 2;
        }
    }
    return 0;
}

or_env *or_env_new(int kind, const int64_t *a, int nargs)
{
    or_env *e = (or_env *)calloc(1, sizeof(or_env));
    e->kind = kind;
    int rc = 0;
    switch (kind) {
    case OR_ENV_ROCK:
        rc = nargs >= 2 ? rock_init(e, (int)a[0], (int)a[1]) : -1;
        e->stochastic = nargs >= 3 ? (int)a[2] : 0;
        e->act_thr = nargs >= 5 ? ((uint64_t)(uint32_t)a[3] | ((uint64_t)(uint32_t)a[4] << 32)) : 0;
        if (e->act_thr == 0) e->act_thr = TAG_MOVE_THR;      /* p_move = .8: the same captured threshold as binomial(1, .8) */
        e->act_gt = nargs >= 6 ? (int)a[5] : 0;
        break;
    case OR_ENV_TAG:
        e->n_opponents = nargs >= 1 ? (int)a[0] : 1;
        e->obs_cells = nargs >= 2 ? (int)a[1] : 29;
        e->move_thr = nargs >= 4 ? ((uint64_t)(uint32_t)a[2] | ((uint64_t)(uint32_t)a[3] << 32)) : 0;
        if (e->move_thr == 0) e->move_thr = TAG_MOVE_THR;
        e->move_gt = nargs >= 5 ? (int)a[4] : 0;
        if (e->n_opponents < 1 || e->n_opponents > MAX_OPP) rc = -1;
        break;
    case OR_ENV_BATTLESHIP:
        if (nargs < 3) { rc = -1; break; }
        e->xs = (int)a[0]; e->ys = (int)a[1]; e->max_len = (int)a[2] + 1;
        /* packed layout limits: <= 4 mask words (cells + 6 spare bits), remaining <= 63 */
        if (e->xs < 1 || e->ys < 1 || e->xs > 16 || e->ys > 16 || e->xs * e->ys > 122) rc = -1;
        if (a[2] < 2 || a[2] > 10) rc = -1;
        break;
    case OR_ENV_TIGER:
        break;
    case OR_ENV_NETWORK:
        rc = nargs >= 2 ? network_init(e, (int)a[0], (int)a[1]) : -1;
        break;
    default:
        rc = -1;
    }
    if (rc) { free(e); return NULL; }
    return e;
}

or_env *or_env_clone(const or_env *e)
{
    or_env *c = (or_env *)malloc(sizeof(or_env));
    memcpy(c, e, sizeof(or_env));
    return c;
}

void or_env_free(or_env *e) { free(e); }

int or_env_n_actions(const or_env *e)
{
    switch (e->kind) {
    case OR_ENV_ROCK: return 5 + e->num_rocks;          /* rock.py:113 */
    case OR_ENV_TAG: return 5;                           /* tag.py:92   */
    case OR_ENV_BATTLESHIP: return e->xs * e->ys;        /* battleship.py:69 */
    case OR_ENV_TIGER: return 3;                         /* tiger.py:52 */
    default: return 2 * e->n_mach + 1;                   /* network.py:33 */
    }
}

int or_env_n_obs(const or_env *e)
{
    switch (e->kind) {
    case OR_ENV_ROCK: return 3;
    case OR_ENV_TAG: return e->obs_cells + 1;            /* tag.py:94 */
    case OR_ENV_BATTLESHIP: return 2;
    default: return 3;
    }
}

int or_env_compact_len(const or_env *e)
{
    switch (e->kind) {
    case OR_ENV_ROCK: return 2 + e->num_rocks;
    case OR_ENV_TAG: return 2 + e->n_opponents;
    case OR_ENV_BATTLESHIP: return 1 + 2 * e->xs * e->ys;
    case OR_ENV_TIGER: return 1;
    default: return e->n_mach;
    }
}

int or_env_reward_kind(const or_env *e)
{
    return (e->kind == OR_ENV_TAG || e->kind == OR_ENV_NETWORK) ? OR_REWARD_F32 : OR_REWARD_I32;
}

/* ------------------------------------------------------------------------ */
/* RockSample                                                                */
/* ------------------------------------------------------------------------ */
/* rock.py:236-241 reset -> 266-271 _get_init_state -> 78-86 Rock.__init__:
 * status = int(np.sign(np.random.uniform(0, 1) - .5)), rocks 0..K-1 in order. */
static int rock_reset(or_env *e, or_ws *np_rng)
{
    e->agent = e->start;
    for (int i = 0; i < e->num_rocks; i++) {
        uint64_t k = or_draw_k53(np_rng);
        e->status[i] = (k > TWO52) - (k < TWO52);
    }
    return 0; /* Obs.NULL */
}

/* rock.py:123-194 step, 401-407 _sample_ob, 383-387 _efficiency,
 * coord.py:79-81 euclidean_distance (ord-1 norm == L1). */
static void rock_step(or_env *e, int action, or_ws *np_rng, int *ob_out, double *rw_out, int *done_out)
{
    int reward = 0, ob = 0;
    /* StochasticRockEnv.step (rock.py:434-504): `if np.random.binomial(1, p=self.p_move):` gates the whole
     * action; _penalization is 0 and the `done = penalization == reward` line is commented out (rock.py:503) */
    const int penal = e->stochastic ? 0 : -100;
    if (e->stochastic && !((or_draw_k53(np_rng) <= e->act_thr) != (e->act_gt != 0))) { *ob_out = 0; *rw_out = 0; *done_out = 0; return; }
    if (action < 4) {
        if (action == 1) {                                   /* EAST  rock.py:135-141 */
            if (e->agent.x + 1 < e->size) e->agent.x += 1;
            else { *ob_out = 0; *rw_out = 10; *done_out = 1; return; }
        } else if (action == 0) {                            /* NORTH rock.py:142-146 */
            if (e->agent.y + 1 < e->size) e->agent.y += 1; else reward = penal;
        } else if (action == 2) {                            /* SOUTH rock.py:147-151 */
            if (e->agent.y - 1 >= 0) e->agent.y -= 1; else reward = penal;
        } else {                                             /* WEST  rock.py:152-156 */
            if (e->agent.x - 1 >= 0) e->agent.x -= 1; else reward = penal;
        }
    }
    if (action == 4) {                                       /* SAMPLE rock.py:160-169 */
        int rock = e->grid[e->agent.x][e->agent.y];
        /* ids >= num_rocks raise IndexError in the reference;
         * the build treats them as "no rock here". */
        if (rock >= 0 && rock < e->num_rocks && e->status[rock] != 0) {
            reward = e->status[rock] == 1 ? 10 : -10;
            e->status[rock] = 0;
        } else reward = penal;
    }
    if (action > 4) {                                        /* CHECK rock.py:171-175 */
        int rock = action - 5;
        int dx = e->agent.x - e->rock_pos[rock].x, dy = e->agent.y - e->rock_pos[rock].y;
        int d = (dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy);
        uint64_t k = or_draw_k53(np_rng);                    /* np.random.binomial(1, eff) */
        int correct = k <= ROCK_THR[d];
        if (correct) ob = e->status[rock] == 1 ? 2 : 1;      /* rock.py:404-407 */
        else ob = e->status[rock] == 1 ? 1 : 2;
    }
    *ob_out = ob; *rw_out = reward;
    *done_out = e->stochastic ? 0 : (reward == -100);        /* rock.py:193 (commented out in the variant, rock.py:503) */
}

/* ------------------------------------------------------------------------ */
/* Tag                                                                       */
/* ------------------------------------------------------------------------ */
/* tag.py:46-50 */
static int tag_inside(coord c)
{
    if (c.y >= 2) return c.x >= 5 && c.x < 8 && c.y < 5;
    return c.x >= 0 && c.x < 10 && c.y >= 0;
}
/* tag.py:52-57 */
static coord tag_coord(int idx)
{
    coord c;
    if (idx < 20) { c.x = idx % 10; c.y = idx / 10; return c; }
    idx -= 20;
    c.x = idx % 3 + 5; c.y = idx / 3 + 2;
    return c;
}
/* tag.py:59-66 */
static int tag_index(coord c)
{
    if (c.y < 2) return c.y * 10 + c.x;
    return 20 + (c.y - 2) * 3 + c.x - 5;
}
/* tag.py:219-226 */
static int tag_sample_ob(const or_env *e, int action)
{
    int ob = tag_index(e->agent);
    if (action < 4)
        for (int i = 0; i < e->n_opponents; i++)
            if (e->opp[i].x == e->agent.x && e->opp[i].y == e->agent.y) ob = e->obs_cells;
    return ob;
}
/* tag.py:97-102 reset, 181-193 _get_init_state, 43-44 TagGrid.sample = randint(0, 29) */
static int tag_reset(or_env *e, or_ws *np_rng)
{
    e->agent = tag_coord((int)or_draw_randint(np_rng, 29));
    for (int i = 0; i < e->n_opponents; i++) e->opp[i] = tag_coord((int)or_draw_randint(np_rng, 29));
    e->num_opp = e->n_opponents;
    return tag_sample_ob(e, 0);
}
/* tag.py:201-207 move_opponent, 260-280 _admissable_actions */
static void tag_move_opponent(or_env *e, int i, or_ws *np_rng)
{
    coord o = e->opp[i], a = e->agent;
    int acts[8], n = 0;                                /* indices into MOVES: N0 E1 S2 W3 */
    if (o.x >= a.x) acts[n++] = 1;
    if (o.y >= a.y) acts[n++] = 0;
    if (o.x <= a.x) acts[n++] = 3;
    if (o.y <= a.y) acts[n++] = 2;
    if (o.x == a.x && o.y > a.y) acts[n++] = 0;
    if (o.y == a.y && o.x > a.x) acts[n++] = 1;
    if (o.x == a.x && o.y < a.y) acts[n++] = 2;
    if (o.y == a.y && o.x < a.x) acts[n++] = 3;
    uint64_t k = or_draw_k53(np_rng);                  /* binomial(1, move_prob) */
    if ((k <= e->move_thr) != (e->move_gt != 0)) {     /* numpy: 1 - [U > q] for p > .5, [U > q] for p <= .5 */
        int pick = acts[or_draw_randint(np_rng, (uint32_t)n)];   /* np.random.choice(actions) */
        coord nx = { o.x + MOVES[pick].x, o.y + MOVES[pick].y };
        if (tag_inside(nx)) e->opp[i] = nx;
    }
}
/* tag.py:108-143 */
static void tag_step(or_env *e, int action, or_ws *np_rng, int *ob_out, double *rw_out, int *done_out)
{
    double reward;
    if (action == 4) {
        int tagged = 0;
        reward = 0.;
        for (int i = 0; i < e->n_opponents; i++) {
            if (e->opp[i].x == e->agent.x && e->opp[i].y == e->agent.y) {
                reward = 10.; tagged = 1; e->num_opp -= 1;
            } else if (tag_inside(e->opp[i]) && e->num_opp > 0) {
                tag_move_opponent(e, i, np_rng);
            }
        }
        if (!tagged) reward = -10.;
    } else {
        reward = -1.;
        coord nx = { e->agent.x + MOVES[action].x, e->agent.y + MOVES[action].y };
        if (tag_inside(nx)) e->agent = nx;
    }
    *ob_out = tag_sample_ob(e, action);
    *rw_out = reward;
    *done_out = (e->num_opp == 0);
}

/* ------------------------------------------------------------------------ */
/* BattleShip                                                                */
/* ------------------------------------------------------------------------ */
/* coord.py:61-62 */
static int bs_inside(const or_env *e, coord c) { return c.x >= 0 && c.y >= 0 && c.x < e->xs && c.y < e->ys; }

/* battleship.py:195-211 */
static int bs_collision(const or_env *e, coord pos, int dir, int length)
{
    for (int i = 0; i < length + 1; i++) {
        coord nx = { pos.x + COMPASS[dir].x, pos.y + COMPASS[dir].y };
        if (!bs_inside(e, nx)) return 1;
        if (e->occ[pos.x][pos.y]) return 1;
        for (int adj = 0; adj < 8; adj++) {
            coord c = { pos.x + COMPASS[adj].x, pos.y + COMPASS[adj].y };
            if (bs_inside(e, c) && e->occ[c.x][c.y]) return 1;
        }
        pos = nx;
    }
    return 0;
}

/* battleship.py:131-137 reset, 167-180 _get_init_state, 182-193 mark_ship,
 * coord.py:68-69 Grid.sample, battleship.py:33-37 Ship.__init__ (pos drawn before direction) */
static int bs_reset(or_env *e, or_ws *np_rng)
{
    memset(e->occ, 0, sizeof(e->occ));
    memset(e->vis, 0, sizeof(e->vis));
    e->remaining = 0;
    for (int length = e->max_len - 1; length >= 2; length--) {
        coord pos; int dir;
        for (;;) {
            int idx = (int)or_draw_randint(np_rng, (uint32_t)(e->xs * e->ys));
            pos.x = idx % e->xs; pos.y = idx / e->xs;          /* coord.py:64-66 */
            dir = (int)or_draw_randint(np_rng, 4);
            if (!bs_collision(e, pos, dir, length)) break;
        }
        for (int i = 0; i < length; i++) {
            e->occ[pos.x][pos.y] = 1;
            e->remaining += 1;                                 /* board is fresh: never visited */
            pos.x += COMPASS[dir].x; pos.y += COMPASS[dir].y;
        }
    }
    return 0;
}

/* The batched build's board contract (include/pomdp_hip.h): a lane holds the board of its NEXT episode
 * too.  Whenever a board is dealt at call counter t — reset() draws it from stream RESET of (lane, t); the auto-reset of
 * a step at t moves the cached board in — the board after it is the reference's reset() (battleship.py:131-137) run on
 * stream NEXT of (lane, t).  bs_deal_next: that second draw; bs_swap_in: the cached board becomes the current one. */
static void bs_deal_next(or_env *e, or_ws *next_rng)
{
    or_env tmp = *e;
    bs_reset(&tmp, next_rng);
    memcpy(e->next_occ, tmp.occ, sizeof(e->occ));
}
static void bs_swap_in(or_env *e)
{
    memcpy(e->occ, e->next_occ, sizeof(e->occ));
    memset(e->vis, 0, sizeof(e->vis));
    e->remaining = 0;
    for (int x = 0; x < e->xs; x++)
        for (int y = 0; y < e->ys; y++) e->remaining += e->occ[x][y];
}

/* battleship.py:91-122 */
static void bs_step(or_env *e, int action, int *ob_out, double *rw_out, int *done_out)
{
    int x = action % e->xs, y = action / e->xs;
    int reward = 0, obs, done = 0;
    if (e->vis[x][y]) { reward -= 10; obs = 0; }
    else {
        if (e->occ[x][y]) { reward -= 1; obs = 1; e->remaining -= 1; }
        else { reward -= 1; obs = 0; }
        e->vis[x][y] = 1;
    }
    if (e->remaining == 0) { reward += e->xs * e->ys; done = 1; }
    *ob_out = obs; *rw_out = reward; *done_out = done;
}

/* ------------------------------------------------------------------------ */
/* Tiger                                                                     */
/* ------------------------------------------------------------------------ */
/* tiger.py:60-66; state_space.sample() draws from the gym-space RNG */
static int tiger_reset(or_env *e, or_ws *space_rng)
{
    e->tiger = (int)or_draw_randint(space_rng, 2);
    return 2; /* Obs.NULL */
}
/* tiger.py:72-88, 117-119, 140-172 */
static void tiger_step(or_env *e, int action, or_ws *np_rng, or_ws *space_rng,
                       int *ob_out, double *rw_out, int *done_out)
{
    int terminal = (action != 2) && (action == e->tiger);
    int rw = action == 2 ? -1 : (terminal ? -20 : 10);
    if (terminal) { *ob_out = e->tiger; *rw_out = rw; *done_out = 1; return; }
    if (action == 0 || action == 1) e->tiger = (int)or_draw_randint(space_rng, 2);
    uint64_t k = or_draw_k53(np_rng);                 /* p = np.random.uniform(), always drawn */
    int ob = 2;
    if (action == 2) {
        int flip = k > TIGER_THR;                     /* p > correct_prob (.85) */
        ob = e->tiger == 0 ? (flip ? 1 : 0) : (flip ? 0 : 1);
    }
    *ob_out = ob; *rw_out = rw; *done_out = 0;
}

/* ------------------------------------------------------------------------ */
/* Network                                                                   */
/* ------------------------------------------------------------------------ */
/* network.py:61-69 */
static int network_reset(or_env *e)
{
    for (int i = 0; i < e->n_mach; i++) e->up[i] = 1;
    return 0; /* Obs.OFF */
}
/* network.py:71-114 */
static void network_step(or_env *e, int action, or_ws *np_rng, int *ob_out, double *rw_out, int *done_out)
{
    int n = e->n_mach, n_fail[MAX_MACH];
    double reward = 0;
    int ob = 2;
    for (int i = 0; i < n; i++) {
        n_fail[i] = 0;
        for (int j = 0; j < e->nb_len[i]; j++) if (e->up[e->nb[i][j]] == 0) n_fail[i] = 1;
    }
    for (int i = 0; i < n; i++) if (e->up[i] == 1) reward += e->nb_len[i] > 2 ? 2 : 1;
    for (int i = 0; i < n; i++) {
        if (e->up[i]) {
            uint64_t k = or_draw_k53(np_rng);
            if (!n_fail[i]) e->up[i] = 1 - (k > NET_FAIL_THR);
            else e->up[i] = 1 - (k > NET_FAILNB_THR);
        }
    }
    if (action < 2 * n) {
        int machine = action / 2, reboot = action % 2;
        if (reboot) {
            reward -= 2.5;
            e->up[machine] = 1;
            ob = or_draw_k53(np_rng) <= NET_OBS_THR;
        } else {
            reward -= .1;
            if (or_draw_k53(np_rng) <= NET_OBS_THR) ob = e->up[machine];
            else ob = 1 - e->up[machine];
        }
    }
    *ob_out = ob; *rw_out = reward; *done_out = 0;
}

/* ------------------------------------------------------------------------ */
/* dispatch                                                                  */
/* ------------------------------------------------------------------------ */
/* The reset that follows a done step inside the step's own call counter.  RockSample / StochasticRock: the rotated pair
 * (layout 3) of the step's SENSOR blocks — stream STEP, blocks b, b + 1 with b = 0 (RockEnv) or 2 (StochasticRockEnv,
 * whose block 0 gates the action) — instead of stream RESET: a step never draws both (a CHECK does not end the episode,
 * rock.py:171-175, 193), so the word is consumed once either way (oracle/philox_ref.py rock_reset_words).  Tag with one
 * opponent: the 5-bit fields of the step's own quad word (a successful TAG draws nothing else; layout 6).  Every other env:
 * stream RESET of (lane, t).  (BattleShip's cached board is the caller's business.) */
void or_ws_philox_auto_reset(or_ws *ws, const or_env *e, uint64_t seed, uint32_t lane, uint64_t t)
{
    if (e->kind == OR_ENV_TAG && e->n_opponents == 1) {                       /* the fields of the step's own quad word */
        or_ws_philox(ws, seed, lane, t, OR_STREAM_RESET);
        ws->layout = 6;
        return;
    }
    if (e->kind == OR_ENV_ROCK) {
        or_ws_philox(ws, seed, lane, t, OR_STREAM_STEP);
        ws->layout = 3; ws->ctr[0] = lane >> 2; ws->blk_base = e->stochastic ? 2u : 0u;
        return;
    }
    or_ws_philox_env(ws, e->kind, seed, lane, t, OR_STREAM_RESET);
}

/* np.random's words inside step() at call counter t */
void or_ws_philox_step(or_ws *ws, const or_env *e, uint64_t seed, uint32_t lane, uint64_t t)
{
    or_ws_philox_env(ws, e->kind, seed, lane, t, OR_STREAM_STEP);
    if (e->kind == OR_ENV_TAG && e->n_opponents == 1) ws->layout = 5;          /* the one-opponent game: the quad's word */
}

int or_env_reset(or_env *e, or_ws *np_rng, or_ws *space_rng)
{
    switch (e->kind) {
    case OR_ENV_ROCK: return rock_reset(e, np_rng);
    case OR_ENV_TAG: return tag_reset(e, np_rng);
    case OR_ENV_BATTLESHIP: return bs_reset(e, np_rng);
    case OR_ENV_TIGER: return tiger_reset(e, space_rng);
    default: return network_reset(e);
    }
}

void or_env_step(or_env *e, int action, or_ws *np_rng, or_ws *space_rng, int *ob, double *reward, int *done)
{
    switch (e->kind) {
    case OR_ENV_ROCK: rock_step(e, action, np_rng, ob, reward, done); break;
    case OR_ENV_TAG: tag_step(e, action, np_rng, ob, reward, done); break;
    case OR_ENV_BATTLESHIP: bs_step(e, action, ob, reward, done); break;
    case OR_ENV_TIGER: tiger_step(e, action, np_rng, space_rng, ob, reward, done); break;
    default: network_step(e, action, np_rng, ob, reward, done); break;
    }
}

/* the reference-side parity format (oracle/ref_harness/harness.py: compact_state) */
void or_env_compact(const or_env *e, int64_t *out)
{
    switch (e->kind) {
    case OR_ENV_ROCK:
        out[0] = e->agent.x; out[1] = e->agent.y;
        for (int i = 0; i < e->num_rocks; i++) out[2 + i] = e->status[i];
        break;
    case OR_ENV_TAG:
        out[0] = tag_index(e->agent);
        for (int i = 0; i < e->n_opponents; i++) out[1 + i] = tag_index(e->opp[i]);
        out[1 + e->n_opponents] = e->num_opp;
        break;
    case OR_ENV_BATTLESHIP: {
        int c = e->xs * e->ys;
        out[0] = e->remaining;
        for (int a = 0; a < c; a++) {
            out[1 + a] = e->occ[a % e->xs][a / e->xs];
            out[1 + c + a] = e->vis[a % e->xs][a / e->xs];
        }
        break;
    }
    case OR_ENV_TIGER:
        out[0] = e->tiger;
        break;
    default:
        for (int i = 0; i < e->n_mach; i++) out[i] = e->up[i];
    }
}

/* ------------------------------------------------------------------------ */
/* packed int32 lane state (the HIP path's HBM layout)    */
/* ------------------------------------------------------------------------ */
static int bs_mask_words(const or_env *e) { return (e->xs * e->ys + 6 + 31) / 32; }

int or_env_words(const or_env *e)
{
    switch (e->kind) {
    case OR_ENV_ROCK: return e->num_rocks <= 12 ? 1 : 2;
    case OR_ENV_BATTLESHIP: return 3 * bs_mask_words(e);   /* occupied, visited (+ remaining), next episode's occupied */
    default: return 1;
    }
}

void or_env_pack(const or_env *e, uint32_t *w)
{
    int nw = or_env_words(e);
    for (int i = 0; i < nw; i++) w[i] = 0;
    switch (e->kind) {
    case OR_ENV_ROCK:
        w[0] = (uint32_t)e->agent.x | ((uint32_t)e->agent.y << 4);
        for (int i = 0; i < e->num_rocks; i++) {
            uint32_t code = (uint32_t)(e->status[i] + 1);
            if (i < 12) w[0] |= code << (8 + 2 * i);
            else w[1] |= code << (2 * (i - 12));
        }
        break;
    case OR_ENV_TAG: {
        w[0] = (uint32_t)tag_index(e->agent);
        for (int i = 0; i < e->n_opponents; i++) w[0] |= (uint32_t)tag_index(e->opp[i]) << (5 + 5 * i);
        int no = e->num_opp < -64 ? -64 : e->num_opp;          /* 7-bit two's complement, saturating */
        w[0] |= ((uint32_t)no & 0x7Fu) << 25;
        break;
    }
    case OR_ENV_BATTLESHIP: {
        int mw = bs_mask_words(e), c = e->xs * e->ys;
        for (int a = 0; a < c; a++) {
            if (e->occ[a % e->xs][a / e->xs]) w[a >> 5] |= 1u << (a & 31);
            if (e->vis[a % e->xs][a / e->xs]) w[mw + (a >> 5)] |= 1u << (a & 31);
            if (e->next_occ[a % e->xs][a / e->xs]) w[2 * mw + (a >> 5)] |= 1u << (a & 31);
        }
        w[2 * mw - 1] |= (uint32_t)e->remaining << 26;
        break;
    }
    case OR_ENV_TIGER:
        w[0] = (uint32_t)e->tiger;
        break;
    default:
        for (int i = 0; i < e->n_mach; i++) w[0] |= (uint32_t)e->up[i] << i;
    }
}

void or_env_unpack(or_env *e, const uint32_t *w)
{
    switch (e->kind) {
    case OR_ENV_ROCK:
        e->agent.x = w[0] & 15; e->agent.y = (w[0] >> 4) & 15;
        for (int i = 0; i < e->num_rocks; i++) {
            uint32_t code = i < 12 ? (w[0] >> (8 + 2 * i)) & 3u : (w[1] >> (2 * (i - 12))) & 3u;
            e->status[i] = (int)code - 1;
        }
        break;
    case OR_ENV_TAG: {
        e->agent = tag_coord((int)(w[0] & 31));
        for (int i = 0; i < e->n_opponents; i++) e->opp[i] = tag_coord((int)((w[0] >> (5 + 5 * i)) & 31));
        int no = (int)((w[0] >> 25) & 0x7F);
        e->num_opp = no >= 64 ? no - 128 : no;
        break;
    }
    case OR_ENV_BATTLESHIP: {
        int mw = bs_mask_words(e), c = e->xs * e->ys;
        for (int a = 0; a < c; a++) {
            e->occ[a % e->xs][a / e->xs] = (w[a >> 5] >> (a & 31)) & 1u;
            e->vis[a % e->xs][a / e->xs] = (w[mw + (a >> 5)] >> (a & 31)) & 1u;
            e->next_occ[a % e->xs][a / e->xs] = (w[2 * mw + (a >> 5)] >> (a & 31)) & 1u;
        }
        e->remaining = (int)(w[2 * mw - 1] >> 26);
        break;
    }
    case OR_ENV_TIGER:
        e->tiger = (int)(w[0] & 1);
        break;
    default:
        for (int i = 0; i < e->n_mach; i++) e->up[i] = (w[0] >> i) & 1u;
    }
}

/* ======================================================================== */
/* mode A driver                                                             */
/* ======================================================================== */
int or_trace_mt(or_env *e, uint32_t seed, uint32_t space_seed, const int64_t *actions, int64_t T,
                int64_t *ob0, int64_t *state0, int64_t *ob, double *reward, uint8_t *done,
                int64_t *state_pre, int64_t *state, int64_t *reset_ob)
{
    or_ws *np_rng = (or_ws *)malloc(sizeof(or_ws)), *sp_rng = (or_ws *)malloc(sizeof(or_ws));
    or_ws_seed_mt(np_rng, seed);          /* env.seed(seed) -> np.random.seed(seed) */
    or_ws_seed_mt(sp_rng, space_seed);    /* gym.spaces RNG (stub-owned, Tiger only) */
    int S = or_env_compact_len(e), nA = or_env_n_actions(e), rc = 0;
    *ob0 = or_env_reset(e, np_rng, sp_rng);
    or_env_compact(e, state0);
    for (int64_t i = 0; i < T; i++) {
        int o, d; double r;
        if (actions[i] < 0 || actions[i] >= nA) { rc = -1; break; }
        or_env_step(e, (int)actions[i], np_rng, sp_rng, &o, &r, &d);
        ob[i] = o; reward[i] = r; done[i] = (uint8_t)d;
        or_env_compact(e, state_pre + i * S);
        reset_ob[i] = -1;
        if (d) reset_ob[i] = or_env_reset(e, np_rng, sp_rng);
        or_env_compact(e, state + i * S);
    }
    free(np_rng); free(sp_rng);
    return rc;
}

/* ======================================================================== */
/* mode B drivers                                                            */
/* ======================================================================== */
int or_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void or_batch_reset(const or_env *proto, uint32_t *state, int32_t *ob, int64_t n,
                    uint64_t seed, uint32_t lane0, uint64_t t, int nthreads)
{
    int W = or_env_words(proto);
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        or_env e = *proto;
        or_ws np_rng, sp_rng;
        uint32_t w[12];
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; i++) {
            or_ws_philox_env(&np_rng, e.kind, seed, lane0 + (uint32_t)i, t, OR_STREAM_RESET);
            or_ws_space(&sp_rng, seed, lane0 + (uint32_t)i, t, OR_STREAM_RESET_SPACE);
            int o = or_env_reset(&e, &np_rng, &sp_rng);
            if (e.kind == OR_ENV_BATTLESHIP) {                 /* ... and the board of the episode after this one */
                or_ws_philox(&np_rng, seed, lane0 + (uint32_t)i, t, OR_STREAM_NEXT);
                bs_deal_next(&e, &np_rng);
            }
            or_env_pack(&e, w);
            for (int j = 0; j < W; j++) state[(int64_t)j * n + i] = w[j];
            if (ob) ob[i] = o;
        }
    }
}

int64_t or_batch_step(const or_env *proto, uint32_t *state, const int32_t *action, int32_t *ob,
                      void *reward, uint8_t *done, int64_t n, uint64_t seed, uint32_t lane0,
                      uint64_t t, int auto_reset, int nthreads)
{
    int W = or_env_words(proto), nA = or_env_n_actions(proto), rk = or_env_reward_kind(proto);
    int64_t bad = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads) reduction(+ : bad)
    {
        or_env e = *proto;
        or_ws np_rng, sp_rng;
        uint32_t w[12];
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; i++) {
            int o = 0, d = 0; double r = 0;
            int a = action[i];
            if (!auto_reset && done[i]) {           /* frozen lane: reference would assert */
                d = 1;
            } else if (a < 0 || a >= nA) {          /* reference: AssertionError; build: no-op + count */
                bad++;
            } else {
                for (int j = 0; j < W; j++) w[j] = state[(int64_t)j * n + i];
                or_env_unpack(&e, w);
                or_ws_philox_step(&np_rng, &e, seed, lane0 + (uint32_t)i, t);
                or_ws_space(&sp_rng, seed, lane0 + (uint32_t)i, t, OR_STREAM_STEP_SPACE);
                or_env_step(&e, a, &np_rng, &sp_rng, &o, &r, &d);
                if (d && auto_reset && e.kind == OR_ENV_BATTLESHIP) {   /* the cached board moves in; the one after it from NEXT */
                    bs_swap_in(&e);
                    or_ws_philox(&np_rng, seed, lane0 + (uint32_t)i, t, OR_STREAM_NEXT);
                    bs_deal_next(&e, &np_rng);
                } else if (d && auto_reset) {
                    or_ws_philox_auto_reset(&np_rng, &e, seed, lane0 + (uint32_t)i, t);
                    or_ws_space(&sp_rng, seed, lane0 + (uint32_t)i, t, OR_STREAM_RESET_SPACE);
                    or_env_reset(&e, &np_rng, &sp_rng);
                }
                or_env_pack(&e, w);
                for (int j = 0; j < W; j++) state[(int64_t)j * n + i] = w[j];
            }
            ob[i] = o;
            if (rk == OR_REWARD_I32) ((int32_t *)reward)[i] = (int32_t)r;
            else ((float *)reward)[i] = (float)r;
            done[i] = (uint8_t)d;
        }
    }
    return bad;
}

void or_batch_compact(const or_env *proto, const uint32_t *state, int64_t *out, int64_t n)
{
    int W = or_env_words(proto), S = or_env_compact_len(proto);
    or_env e = *proto;
    uint32_t w[12];
    for (int64_t i = 0; i < n; i++) {
        for (int j = 0; j < W; j++) w[j] = state[(int64_t)j * n + i];
        or_env_unpack(&e, w);
        or_env_compact(&e, out + i * S);
    }
}

/* the bench's synthetic uniform policy (oracle/philox_ref.py: synthetic_actions) */
void or_synthetic_actions(int32_t *action, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t,
                          uint32_t n_actions, int nthreads)
{
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t i = 0; i < n; i++) {
        uint32_t lane = lane0 + (uint32_t)i;
        uint32_t c[4] = { lane >> 2, (uint32_t)t, (uint32_t)(t >> 32), (uint32_t)OR_STREAM_ACTION << 24 }, o[4];
        or_philox4x32_10(c, key, o);
        action[i] = (int32_t)(((uint64_t)o[lane & 3u] * n_actions) >> 32);
    }
}

/* bench.py's cpu_baseline: `steps` steps of n lanes under the synthetic policy with auto-reset — the workload of the GPU's
 * timed region.  Lanes are independent, so every thread owns a contiguous chunk of lanes for the whole run (static split,
 * its pages first touched by itself) and walks it in blocks of lanes that stay in its cache across the steps of a block
 * pass: no barrier and no shared cache line between threads anywhere in the timed part, which is what lets the loop scale
 * with the cores it is given.  Same results as or_batch_reset + steps x (or_synthetic_actions, or_batch_step): the draws of
 * a lane depend on (seed, lane, t) only. */
double or_bench_loop(const or_env *proto, int64_t n, int64_t steps, uint64_t seed, int nthreads, int64_t *n_done)
{
    int W = or_env_words(proto), nA = or_env_n_actions(proto);
    const uint64_t aseed = seed ^ 0x5DEECE66DULL;
    const uint32_t akey[2] = { (uint32_t)aseed, (uint32_t)(aseed >> 32) };
    uint32_t *state = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)W * (size_t)n);
    int64_t dsum = 0;
    double el = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads) reduction(+ : dsum)
    {
        or_env e = *proto;
        or_ws np_rng, sp_rng;
        uint32_t w[12];
        const int nt = omp_get_num_threads(), tid = omp_get_thread_num();
        const int64_t lo = n * tid / nt, hi = n * (tid + 1) / nt;
        for (int64_t i = lo; i < hi; i++) {                       /* reset at t = 0 (first touch of the thread's own chunk) */
            or_ws_philox_env(&np_rng, e.kind, seed, (uint32_t)i, 0, OR_STREAM_RESET);
            or_ws_space(&sp_rng, seed, (uint32_t)i, 0, OR_STREAM_RESET_SPACE);
            or_env_reset(&e, &np_rng, &sp_rng);
            if (e.kind == OR_ENV_BATTLESHIP) { or_ws_philox(&np_rng, seed, (uint32_t)i, 0, OR_STREAM_NEXT); bs_deal_next(&e, &np_rng); }
            or_env_pack(&e, w);
            for (int j = 0; j < W; j++) state[(int64_t)j * n + i] = w[j];
        }
#pragma omp barrier
        double t0 = omp_get_wtime();
        for (int64_t i = lo; i < hi; i++) {
            for (int j = 0; j < W; j++) w[j] = state[(int64_t)j * n + i];
            or_env_unpack(&e, w);
            for (int64_t s = 1; s <= steps; s++) {
                uint32_t c[4] = { (uint32_t)i >> 2, (uint32_t)s, (uint32_t)((uint64_t)s >> 32), (uint32_t)OR_STREAM_ACTION << 24 }, o4[4];
                or_philox4x32_10(c, akey, o4);
                const int a = (int)(((uint64_t)o4[i & 3] * (uint32_t)nA) >> 32);
                int o, d; double r;
                or_ws_philox_step(&np_rng, &e, seed, (uint32_t)i, (uint64_t)s);
                or_ws_space(&sp_rng, seed, (uint32_t)i, (uint64_t)s, OR_STREAM_STEP_SPACE);
                or_env_step(&e, a, &np_rng, &sp_rng, &o, &r, &d);
                if (d && e.kind == OR_ENV_BATTLESHIP) {
                    bs_swap_in(&e);
                    or_ws_philox(&np_rng, seed, (uint32_t)i, (uint64_t)s, OR_STREAM_NEXT);
                    bs_deal_next(&e, &np_rng);
                } else if (d) {
                    or_ws_philox_auto_reset(&np_rng, &e, seed, (uint32_t)i, (uint64_t)s);
                    or_ws_space(&sp_rng, seed, (uint32_t)i, (uint64_t)s, OR_STREAM_RESET_SPACE);
                    or_env_reset(&e, &np_rng, &sp_rng);
                }
                dsum += d;
            }
            or_env_pack(&e, w);
            for (int j = 0; j < W; j++) state[(int64_t)j * n + i] = w[j];
        }
#pragma omp barrier
        if (tid == 0) el = omp_get_wtime() - t0;
    }
    if (n_done) *n_done = dsum;
    free(state);
    return el;
}

/* The reduction the reference's callers apply to the stream of step() results under a random policy — `r += discount * rw;
 * discount *= .95` per step, `eps.append(r)` per episode, `sum(eps)` over them (network.py:175-191; rock.py:553-575 has the
 * same two lines with the factors swapped) — for every lane, lane-major, over the k steps of or_synthetic_actions +
 * or_batch_step(auto_reset) at call counters t0 .. t0 + k - 1 (`actions`, int32 [k][n], replaces the policy when not NULL: the
 * fixtures' tapes).  acc: double [4][pitch] = ret, disc, ret_done, ret_sum; cnt:
 * int32 [2][pitch] = episodes, steps; all in/out (include/pomdp_hip.h: pomdp_collect_returns).  The reward is or_env_step's
 * double — the reference's own value (Network: base - .1, not its float32).  Returns the number of done steps. */
int64_t or_batch_collect_returns(const or_env *proto, uint32_t *state, double *acc, int32_t *cnt, int64_t pitch, double discount,
                                 const int32_t *actions, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t0, int64_t k,
                                 int nthreads)
{
    int W = or_env_words(proto), nA = or_env_n_actions(proto);
    const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    int64_t dsum = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads) reduction(+ : dsum)
    {
        or_env e = *proto;
        or_ws np_rng, sp_rng;
        uint32_t w[12];
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; i++) {
            const uint32_t lane = lane0 + (uint32_t)i;
            double ret = acc[i], disc = acc[pitch + i], ret_done = acc[2 * pitch + i], ret_sum = acc[3 * pitch + i];
            int32_t episodes = cnt[i];
            for (int j = 0; j < W; j++) w[j] = state[(int64_t)j * n + i];
            or_env_unpack(&e, w);
            for (int64_t s = 0; s < k; s++) {
                const uint64_t t = t0 + (uint64_t)s;
                uint32_t c[4] = { lane >> 2, (uint32_t)t, (uint32_t)(t >> 32), (uint32_t)OR_STREAM_ACTION << 24 }, o4[4];
                or_philox4x32_10(c, key, o4);
                const int a = actions ? actions[s * n + i] : (int)(((uint64_t)o4[lane & 3u] * (uint32_t)nA) >> 32);
                int o, d; double r;
                if (a < 0 || a >= nA) {                     /* a tape's out-of-range action (the reference asserts): the lane is left
                                                               untouched — a step that returns (0, 0, 0) — as in or_batch_step */
                    o = 0; d = 0; r = 0.0;
                } else {
                    or_ws_philox_step(&np_rng, &e, seed, lane, t);
                    or_ws_space(&sp_rng, seed, lane, t, OR_STREAM_STEP_SPACE);
                    or_env_step(&e, a, &np_rng, &sp_rng, &o, &r, &d);
                }
                {   /* r += discount * rw; discount *= .95 (network.py:186-187) — separate multiply and add */
                    const double term = disc * r;
                    ret = ret + term;
                    disc = disc * discount;
                }
                if (d) {                                    /* eps.append(r) (network.py:188); the next episode starts at 0 / 1 */
                    ret_done = ret;
                    ret_sum = ret_sum + ret;
                    episodes++;
                    ret = 0.0; disc = 1.0;
                    dsum++;
                    if (e.kind == OR_ENV_BATTLESHIP) {
                        bs_swap_in(&e);
                        or_ws_philox(&np_rng, seed, lane, t, OR_STREAM_NEXT);
                        bs_deal_next(&e, &np_rng);
                    } else {
                        or_ws_philox_auto_reset(&np_rng, &e, seed, lane, t);
                        or_ws_space(&sp_rng, seed, lane, t, OR_STREAM_RESET_SPACE);
                        or_env_reset(&e, &np_rng, &sp_rng);
                    }
                }
            }
            or_env_pack(&e, w);
            for (int j = 0; j < W; j++) state[(int64_t)j * n + i] = w[j];
            acc[i] = ret; acc[pitch + i] = disc; acc[2 * pitch + i] = ret_done; acc[3 * pitch + i] = ret_sum;
            cnt[i] = episodes; cnt[pitch + i] += (int32_t)k;
        }
    }
    return dsum;
}

/* ======================================================================== */
/* planner hooks: _generate_legal and random rollouts                        */
/* ======================================================================== */
int or_env_legal(const or_env *e, int *list)
{
    int n = 0;
    switch (e->kind) {
    case OR_ENV_ROCK: {                                   /* rock.py:273-291 */
        list[n++] = 1;                                    /* EAST is always legal */
        if (e->agent.y + 1 < e->size) list[n++] = 0;      /* NORTH */
        if (e->agent.y - 1 >= 0) list[n++] = 2;           /* SOUTH */
        if (e->agent.x - 1 >= 0) list[n++] = 3;           /* WEST */
        int rock = e->grid[e->agent.x][e->agent.y];
        if (rock >= 0 && rock < e->num_rocks && e->status[rock] != 0) list[n++] = 4;   /* SAMPLE */
        for (int i = 0; i < e->num_rocks; i++)            /* CHECK grid[rock.pos] for every uncollected rock */
            if (e->status[i] != 0) list[n++] = e->grid[e->rock_pos[i].x][e->rock_pos[i].y] + 1 + 4;
        break;
    }
    case OR_ENV_BATTLESHIP:                               /* battleship.py:157-165: unvisited cells */
        for (int a = 0; a < e->xs * e->ys; a++)
            if (!e->vis[a % e->xs][a / e->xs]) list[n++] = a;
        break;
    default: {                                            /* tag.py:228-229, tiger.py:111-112, network.py:130-131 */
        int na = or_env_n_actions(e);
        for (int a = 0; a < na; a++) list[n++] = a;
    }
    }
    return n;
}

void or_batch_legal(const or_env *proto, const uint32_t *state, int32_t *out, int32_t *len, int64_t n)
{
    int W = or_env_words(proto);
    or_env e = *proto;
    uint32_t w[12];
    int list[OR_MAX_LEGAL];
    for (int64_t i = 0; i < n; i++) {
        for (int j = 0; j < W; j++) w[j] = state[(int64_t)j * n + i];
        or_env_unpack(&e, w);
        int l = or_env_legal(&e, list);
        len[i] = l;
        for (int j = 0; j < OR_MAX_LEGAL; j++) out[i * OR_MAX_LEGAL + j] = j < l ? list[j] : -1;
    }
}

void or_batch_rollout(const or_env *proto, const uint32_t *state, int64_t n_roots, int64_t sims_per_root,
                      int depth, double discount, int policy_all_actions, uint64_t seed, uint32_t lane0,
                      uint64_t t0, double *ret, int32_t *n_steps, int32_t *first_action, int32_t *last_ob,
                      uint8_t *terminated, int nthreads)
{
    int W = or_env_words(proto), nA = or_env_n_actions(proto);
    int64_t n = n_roots * sims_per_root;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        or_env e = *proto;
        or_ws np_rng, sp_rng, pol;
        uint32_t w[12];
        int list[OR_MAX_LEGAL];
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; i++) {
            int64_t root = i / sims_per_root;
            for (int j = 0; j < W; j++) w[j] = state[(int64_t)j * n_roots + root];
            or_env_unpack(&e, w);
            uint32_t lane = lane0 + (uint32_t)i;
            double r_acc = 0.0, disc = 1.0;
            int k = 0, d = 0, o = 0, first = -1;
            for (; k < depth && !d; k++) {
                int len;
                if (policy_all_actions) { len = nA; for (int a = 0; a < nA; a++) list[a] = a; }
                else len = or_env_legal(&e, list);
                if (len == 0) break;                       /* BattleShip with every cell visited */
                if (k == 0) or_ws_philox(&pol, seed, lane, t0, OR_STREAM_ROLLOUT);   /* word k of this stream picks step k */
                int a = list[((uint64_t)or_ws_next32(&pol) * (uint64_t)len) >> 32];
                if (k == 0) first = a;
                or_ws_philox_step(&np_rng, &e, seed, lane, t0 + (uint64_t)k);
                or_ws_space(&sp_rng, seed, lane, t0 + (uint64_t)k, OR_STREAM_STEP_SPACE);
                double r;
                or_env_step(&e, a, &np_rng, &sp_rng, &o, &r, &d);
                double term = disc * r;
                r_acc = r_acc + term;
                disc = disc * discount;
            }
            ret[i] = r_acc; n_steps[i] = k; first_action[i] = first; last_ob[i] = o; terminated[i] = (uint8_t)d;
        }
    }
}

/* The caller's reduction of those simulations to action values (pomdp_oracle.h: or_plan_reduce). */
void or_plan_reduce(const double *ret, const int32_t *first_action, int64_t n_roots, int64_t sims_per_root, int n_actions,
                    int stride, double *q, int32_t *visits, int32_t *best, double *value)
{
    for (int64_t r = 0; r < n_roots; r++) {
        const double *rr = ret + r * sims_per_root;
        const int32_t *fa = first_action + r * sims_per_root;
        int b = -1;
        double bq = 0.0;
        for (int a = 0; a < n_actions; a++) {
            double total = 0.0;
            int32_t cnt = 0;
            for (int64_t c0 = 0; c0 < sims_per_root; c0 += OR_PLAN_CHUNK) {
                double part = 0.0;
                int64_t c1 = c0 + OR_PLAN_CHUNK < sims_per_root ? c0 + OR_PLAN_CHUNK : sims_per_root;
                for (int64_t j = c0; j < c1; j++)
                    if (fa[j] == a) { part = part + rr[j]; cnt++; }
                total = total + part;
            }
            double qa = cnt > 0 ? total / (double)cnt : 0.0;
            q[r * stride + a] = qa;
            visits[r * stride + a] = cnt;
            if (cnt > 0 && (b < 0 || qa > bq)) { b = a; bq = qa; }
        }
        best[r] = b;
        if (value) value[r] = b >= 0 ? bq : 0.0;
    }
}

/* ======================================================================== */
/* planner hook: observation likelihood _compute_prob(action, next_state, ob) */
/* ======================================================================== */
/* rock.py:383-387 eff(d) = (1 + pow(2, -d / 20)) * .5 for L1 distance d, captured from the reference
 * (tests/golden/thresholds.json: rock_eff_hex) */
static const double ROCK_EFF[29] = {
    0x1.0000000000000p+0, 0x1.f7479a6ec0218p-1, 0x1.eedb4008bd589p-1, 0x1.e6b859ae6b1b6p-1,
    0x1.dedc66d6df090p-1, 0x1.d744fccad69d6p-1, 0x1.cfefc5e67299fp-1, 0x1.c8da80e16d9f0p-1,
    0x1.c203001d9572ep-1, 0x1.bb6728fb505dcp-1, 0x1.b504f333f9de6p-1, 0x1.aeda6839e3c90p-1,
    0x1.a8e5a29dca9b6p-1, 0x1.a324cd798d804p-1, 0x1.9d9623dffc194p-1, 0x1.9837f0518db8ap-1,
    0x1.93088c35d733ap-1, 0x1.8e065f5995efcp-1, 0x1.892fdf7128332p-1, 0x1.84838f9f4c1d6p-1,
    0x1.8000000000000p-1, 0x1.7ba3cd376010cp-1, 0x1.776da0045eac4p-1, 0x1.735c2cd7358dbp-1,
    0x1.6f6e336b6f848p-1, 0x1.6ba27e656b4ebp-1, 0x1.67f7e2f3394cfp-1, 0x1.646d4070b6cf8p-1,
    0x1.6101800ecab97p-1 };

double or_env_compute_prob(const or_env *e, int action, int ob)
{
    switch (e->kind) {
    case OR_ENV_ROCK: {                                        /* rock.py:250-264 */
        if (action <= 4) return ob == 0;
        int rock = action - 5;
        int dx = e->agent.x - e->rock_pos[rock].x, dy = e->agent.y - e->rock_pos[rock].y;
        double eff = ROCK_EFF[(dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy)];
        if (ob == 2 && e->status[rock] == 1) return eff;
        if (ob == 1 && e->status[rock] == -1) return eff;
        return 1 - eff;
    }
    case OR_ENV_TAG: {                                         /* tag.py:209-217 */
        int p_ob = ob == tag_index(e->agent);
        if (ob == e->obs_cells)
            for (int i = 0; i < e->n_opponents; i++)
                if (e->opp[i].x == e->agent.x && e->opp[i].y == e->agent.y) return 1.;
        return p_ob;
    }
    case OR_ENV_BATTLESHIP: {                                  /* battleship.py:80-89: reads the live grid */
        int x = action % e->xs, y = action / e->xs;
        if (ob == 0 && e->vis[x][y]) return 1;
        if (ob == 1 && e->occ[x][y]) return 1;
        return ob == 0;
    }
    case OR_ENV_TIGER: {                                       /* tiger.py:125-138 */
        double p_ob = 0.0;
        if (action == 2 && ob != 2) p_ob = (e->tiger == ob) ? .85 : 1 - .85;
        else if (action != 2 && ob == 2) p_ob = 1.;
        return p_ob;
    }
    default:                                                   /* network.py:43-55 */
        if (action < e->n_mach * 2) return e->up[action / 2] == ob ? .95 : 1 - .95;
        if (ob == 2) return 1.;
        return 0;
    }
}

void or_batch_compute_prob(const or_env *proto, const uint32_t *state, const int32_t *action, const int32_t *ob,
                           double *out, int64_t n)
{
    int W = or_env_words(proto);
    or_env e = *proto;
    uint32_t w[12];
    for (int64_t i = 0; i < n; i++) {
        for (int j = 0; j < W; j++) w[j] = state[(int64_t)j * n + i];
        or_env_unpack(&e, w);
        out[i] = or_env_compute_prob(&e, action[i], ob[i]);
    }
}

/* ======================================================================== */
/* heuristic-policy support: side statistics, history sums, preferred lists  */
/* ======================================================================== */
/* fresh Rock objects for one lane (rock.py:81-86) */
static void belief_fresh_lane(const or_rock_belief *b, int K, int64_t i, int64_t n)
{
    for (int j = 0; j < K; j++) {
        int64_t k = (int64_t)j * n + i;
        b->count[k] = 0; b->measured[k] = 0; b->lkv[k] = 1.; b->lkw[k] = 1.; b->prob_valuable[k] = .5;
    }
}

void or_batch_rock_belief_reset(const or_env *proto, const or_rock_belief *b, const uint8_t *where, int64_t n)
{
    for (int64_t i = 0; i < n; i++) {
        if (where && !where[i]) continue;
        belief_fresh_lane(b, proto->num_rocks, i, n);
    }
}

/* one lane of or_batch_rock_belief_update: `e` is the lane's unpacked stored (post auto-reset) state */
static void belief_update_lane(const or_env *e, const or_rock_belief *b, int64_t i, int64_t n, int a, int ob, int done,
                               int auto_reset)
{
    int K = e->num_rocks;
    if (done) {                                               /* reset() builds new Rock objects */
        if (auto_reset) belief_fresh_lane(b, K, i, n);
        return;
    }
    if (a <= 4 || a >= 5 + K || ob == 0) return;              /* not an executed CHECK */
    int rock = a - 5;
    int64_t k = (int64_t)rock * n + i;
    int dx = e->agent.x - e->rock_pos[rock].x, dy = e->agent.y - e->rock_pos[rock].y;
    double eff = ROCK_EFF[(dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy)];          /* rock.py:180 */
    b->measured[k] += 1;                                      /* rock.py:178 */
    if (ob == 2) {                                            /* rock.py:182-185 */
        b->count[k] += 1; b->lkv[k] *= eff; b->lkw[k] *= (1 - eff);
    } else {                                                  /* rock.py:186-189 */
        b->count[k] -= 1; b->lkw[k] *= eff; b->lkv[k] *= (1 - eff);
    }
    double denom = (.5 * b->lkv[k]) + (.5 * b->lkw[k]);       /* rock.py:190-191 */
    b->prob_valuable[k] = (.5 * b->lkv[k]) / denom;
}

void or_batch_rock_belief_update(const or_env *proto, const uint32_t *state, const int32_t *action, const int32_t *ob,
                                 const uint8_t *done, int auto_reset, const or_rock_belief *b, int64_t n)
{
    int W = or_env_words(proto);
    or_env e = *proto;
    uint32_t w[12];
    for (int64_t i = 0; i < n; i++) {
        for (int j = 0; j < W; j++) w[j] = state[(int64_t)j * n + i];
        or_env_unpack(&e, w);
        belief_update_lane(&e, b, i, n, action[i], ob[i], done[i], auto_reset);
    }
}

static void history_clear_lane(const or_history *h, int K, int64_t i, int64_t n)
{
    h->size[i] = 0; h->last_action[i] = -1; h->last_ob[i] = -1;
    for (int j = 0; j < K; j++) { h->total_sample[(int64_t)j * n + i] = 0; h->total_move[(int64_t)j * n + i] = 0; }
}

void or_batch_history_clear(const or_env *proto, const or_history *h, const uint8_t *where, int64_t n)
{
    int K = proto->kind == OR_ENV_ROCK ? proto->num_rocks : 0;
    for (int64_t i = 0; i < n; i++) {
        if (where && !where[i]) continue;
        history_clear_lane(h, K, i, n);
    }
}

/* one lane of or_batch_history_append */
static void history_append_lane(const or_history *h, int K, int64_t i, int64_t n, int observation, int a, int o, int done,
                                int auto_reset)
{
    if (done && auto_reset) {                                 /* next episode: a new, empty History */
        history_clear_lane(h, K, i, n);
        return;
    }
    if (h->max_size >= 0) {                                   /* rock.py:541-544: the records themselves */
        int sz = h->size[i];
        if (sz > h->max_size) {                               /* self._history.pop(0) */
            for (int r = 1; r < sz; r++) {
                h->rec_obs[(int64_t)(r - 1) * n + i] = h->rec_obs[(int64_t)r * n + i];
                h->rec_act[(int64_t)(r - 1) * n + i] = h->rec_act[(int64_t)r * n + i];
                h->rec_next[(int64_t)(r - 1) * n + i] = h->rec_next[(int64_t)r * n + i];
            }
            sz -= 1;
        }
        h->rec_obs[(int64_t)sz * n + i] = observation;        /* self._history.append(transition) */
        h->rec_act[(int64_t)sz * n + i] = a;
        h->rec_next[(int64_t)sz * n + i] = o;
        h->size[i] = sz + 1; h->last_action[i] = a; h->last_ob[i] = o;
        return;
    }
    h->size[i] += 1; h->last_action[i] = a; h->last_ob[i] = o;
    if (a >= 5 && a < 5 + K) {
        int64_t k = (int64_t)(a - 5) * n + i;
        if (o == 2) h->total_sample[k] += 1;                  /* rock.py:305-309 */
        else if (o == 1) h->total_sample[k] -= 1;
        if (o == 2) h->total_move[k] += 1;                    /* rock.py:329-333: elif on transition.observation */
        else if (observation == 1) h->total_move[k] -= 1;
    }
}

void or_batch_history_append(const or_env *proto, const or_history *h, const int32_t *observation,
                             const int32_t *action, const int32_t *next_observation, const uint8_t *done,
                             int auto_reset, int64_t n)
{
    int K = proto->kind == OR_ENV_ROCK ? proto->num_rocks : 0;
    for (int64_t i = 0; i < n; i++)
        history_append_lane(h, K, i, n, observation[i], action[i], next_observation[i], done[i], auto_reset);
}

/* the two per-rock sums over the history's transitions: rock.py:303-310 (sample) and rock.py:327-334 (move) */
static int history_total(const or_history *h, int rock, int move, int64_t i, int64_t n)
{
    if (h->max_size < 0) return (move ? h->total_move : h->total_sample)[(int64_t)rock * n + i];
    int total = 0;
    for (int r = 0; r < h->size[i]; r++) {                        /* for transition in history */
        if (h->rec_act[(int64_t)r * n + i] != rock + 5) continue;
        int nx = h->rec_next[(int64_t)r * n + i];
        if (nx == 2) total += 1;
        else if ((move ? h->rec_obs[(int64_t)r * n + i] : nx) == 1) total -= 1;
    }
    return total;
}

/* rock.py:293-374 */
static int rock_preferred(const or_env *e, const or_rock_belief *b, const or_history *h, int64_t i, int64_t n, int *list)
{
    int K = e->num_rocks, cnt = 0;
    int rock = e->grid[e->agent.x][e->agent.y];
    /* ids >= num_rocks raise IndexError in the reference; the build treats such a cell as empty */
    if (rock >= 0 && rock < K && e->status[rock] != 0 && h->size[i]) {
        if (history_total(h, rock, 0, i, n) > 0) { list[0] = 4; return 1; }                    /* rock.py:311-313 */
    }
    int all_bad = 1, north = 0, south = 0, west = 0, east = 0;
    for (int idx = 0; idx < K; idx++) {
        if (e->status[idx] == 0) continue;
        if (history_total(h, idx, 1, i, n) >= 0) {                                              /* rock.py:335-345 */
            all_bad = 0;
            if (e->rock_pos[idx].y > e->agent.y) north = 1;
            else if (e->rock_pos[idx].y < e->agent.y) south = 1;
            else if (e->rock_pos[idx].x < e->agent.x) west = 1;
            else if (e->rock_pos[idx].x > e->agent.x) east = 1;
        }
    }
    if (all_bad) { list[0] = 1; return 1; }                                                     /* rock.py:347-349 */
    if (e->agent.y + 1 < e->size && north) list[cnt++] = 0;                                     /* rock.py:358-368 */
    if (east) list[cnt++] = 1;
    if (e->agent.y - 1 >= 0 && south) list[cnt++] = 2;
    if (e->agent.x - 1 >= 0 && west) list[cnt++] = 3;
    for (int idx = 0; idx < K; idx++) {                                                         /* rock.py:370-372 */
        int64_t k = (int64_t)idx * n + i;
        int c = b->count[k];
        if (e->status[idx] != 0 && b->measured[k] < 5 && (c < 0 ? -c : c) < 2 && 0 < b->prob_valuable[k] &&
            b->prob_valuable[k] < 1)
            list[cnt++] = idx + 5;
    }
    if (cnt == 0) return or_env_legal(e, list);                                                 /* rock.py:374-375 */
    return cnt;
}

/* tag.py:231-243, 68-74 is_corner, coord.py:75-77 opposite */
static int tag_preferred(const or_env *e, const or_history *h, int64_t i, int *list)
{
    int cnt = 0;
    if (h->size[i] == 0) return or_env_legal(e, list);
    int corner = tag_inside(e->agent) &&
                 (e->agent.y < 2 ? (e->agent.x == 0 || e->agent.x == 9) : (e->agent.y == 4 && (e->agent.x == 5 || e->agent.x == 7)));
    if (h->last_ob[i] == 29 && corner) { list[0] = 4; return 1; }   /* grid.n_tiles == 29 whatever obs_cells is */
    for (int d = 0; d < 4; d++) {
        coord c = { e->agent.x + MOVES[d].x, e->agent.y + MOVES[d].y };
        if (h->last_action[i] != (d + 2) % 4 && tag_inside(c)) list[cnt++] = d;
    }
    return cnt;
}

/* `_generate_preferred(history)` of one lane whose state is unpacked in `e` */
static int preferred_lane(const or_env *e, const or_rock_belief *b, const or_history *h, int64_t i, int64_t n, int *list)
{
    if (e->kind == OR_ENV_ROCK) return rock_preferred(e, b, h, i, n, list);
    if (e->kind == OR_ENV_TAG) return tag_preferred(e, h, i, list);
    return or_env_legal(e, list);
}

void or_batch_preferred(const or_env *proto, const uint32_t *state, const or_rock_belief *b, const or_history *h,
                        int32_t *out, int32_t *len, int64_t n)
{
    int W = or_env_words(proto);
    or_env e = *proto;
    uint32_t w[12];
    int list[OR_MAX_LEGAL];
    for (int64_t i = 0; i < n; i++) {
        for (int j = 0; j < W; j++) w[j] = state[(int64_t)j * n + i];
        or_env_unpack(&e, w);
        int l = preferred_lane(&e, b, h, i, n, list);
        len[i] = l;
        for (int j = 0; j < OR_MAX_LEGAL; j++) out[i * OR_MAX_LEGAL + j] = j < l ? list[j] : -1;
    }
}

void or_batch_rock_select_target(const or_env *proto, const uint32_t *state, const or_rock_belief *b,
                                 int32_t *target, int64_t n)
{
    int W = or_env_words(proto);
    or_env e = *proto;
    uint32_t w[12];
    for (int64_t i = 0; i < n; i++) {
        for (int j = 0; j < W; j++) w[j] = state[(int64_t)j * n + i];
        or_env_unpack(&e, w);
        double best = e.size * 2;                                                  /* rock.py:391 */
        int best_rock = -1;
        for (int idx = 0; idx < e.num_rocks; idx++)
            if (e.status[idx] != 0 && b->count[(int64_t)idx * n + i] >= 0) {
                double dx = e.agent.x - e.rock_pos[idx].x, dy = e.agent.y - e.rock_pos[idx].y;
                double d = sqrt(dx * dx + dy * dy);                                /* coord.py:83-85 */
                if (d < best) { best = d; best_rock = idx; }
            }
        target[i] = best_rock;
    }
}

void or_batch_pick(const int32_t *list, const int32_t *len, int stride, int32_t *action, int64_t n, uint64_t seed,
                   uint32_t lane0, uint64_t t)
{
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    for (int64_t i = 0; i < n; i++) {
        uint32_t lane = lane0 + (uint32_t)i;
        uint32_t c[4] = { lane >> 2, (uint32_t)t, (uint32_t)(t >> 32), (uint32_t)OR_STREAM_ACTION << 24 }, o[4];
        or_philox4x32_10(c, key, o);
        action[i] = len[i] > 0 ? list[i * stride + (int32_t)(((uint64_t)o[lane & 3u] * (uint32_t)len[i]) >> 32)] : -1;
    }
}

/* The reference's heuristic rollout loop (rock.py:557-573) for a batch, k steps of every lane in one call — what
 * pomdp_heuristic_steps fuses on the device — composed from the SAME per-lane pieces the per-step batch functions
 * above are made of (preferred_lane, the pick of or_batch_pick, the step + auto-reset of or_batch_step,
 * belief_update_lane, history_append_lane), lane-major so that no [n][OR_MAX_LEGAL] list array is ever materialised
 * (tests/test_oracle_golden.py checks it against the per-step call sequence).  Step s of a lane, at call counter t0 + s:
 *   list = _generate_preferred(history); a = list[(w * len) >> 32], w = the synthetic policy's word of (seed, lane, t)
 *   (ob, reward, done) = step(a), auto-reset as in or_batch_step; side statistics; history.append(Transition(prev_ob,
 *   a, reward, ob, done)); prev_ob <- ob, or what reset() returned on a lane that auto-reset.
 * Outputs: rows [k][n] of action / ob / reward / done; state, b, h, prev_ob are updated in place.  Without auto_reset,
 * done_in (uint8 [n], may be NULL) marks the lanes that are already frozen; frozen lane-steps write (-1, 0, 0, 1). */
void or_batch_heuristic_steps(const or_env *proto, uint32_t *state, const or_rock_belief *b, const or_history *h,
                              int32_t *prev_ob, const uint8_t *done_in, int32_t *action, int32_t *ob, void *reward,
                              uint8_t *done, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t0, int64_t k,
                              int auto_reset, int nthreads)
{
    int W = or_env_words(proto), nA = or_env_n_actions(proto), rk = or_env_reward_kind(proto);
    int K = proto->kind == OR_ENV_ROCK ? proto->num_rocks : 0;
    const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        or_env e = *proto;
        or_ws np_rng, sp_rng;
        uint32_t w[12];
        int list[OR_MAX_LEGAL];
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; i++) {
            const uint32_t lane = lane0 + (uint32_t)i;
            int was_done = (!auto_reset && done_in) ? done_in[i] != 0 : 0;
            for (int j = 0; j < W; j++) w[j] = state[(int64_t)j * n + i];
            or_env_unpack(&e, w);
            int pob = prev_ob[i];
            for (int64_t s = 0; s < k; s++) {
                const uint64_t t = t0 + (uint64_t)s;
                const int64_t row = s * n + i;
                int a = -1, o = 0, d = 0; double r = 0;
                if (was_done) {
                    d = 1;
                } else {
                    int l = preferred_lane(&e, b, h, i, n, list);
                    uint32_t c[4] = { lane >> 2, (uint32_t)t, (uint32_t)(t >> 32), (uint32_t)OR_STREAM_ACTION << 24 }, o4[4];
                    or_philox4x32_10(c, key, o4);
                    a = l > 0 ? list[(int32_t)(((uint64_t)o4[lane & 3u] * (uint32_t)l) >> 32)] : -1;
                    int fresh_ob = 0;
                    if (a >= 0 && a < nA) {
                        or_ws_philox_step(&np_rng, &e, seed, lane, t);
                        or_ws_space(&sp_rng, seed, lane, t, OR_STREAM_STEP_SPACE);
                        or_env_step(&e, a, &np_rng, &sp_rng, &o, &r, &d);
                        if (d && auto_reset && e.kind == OR_ENV_BATTLESHIP) {
                            bs_swap_in(&e);
                            or_ws_philox(&np_rng, seed, lane, t, OR_STREAM_NEXT);
                            bs_deal_next(&e, &np_rng);
                        } else if (d && auto_reset) {
                            or_ws_philox_auto_reset(&np_rng, &e, seed, lane, t);
                            or_ws_space(&sp_rng, seed, lane, t, OR_STREAM_RESET_SPACE);
                            fresh_ob = or_env_reset(&e, &np_rng, &sp_rng);
                        }
                    }
                    if (K) belief_update_lane(&e, b, i, n, a, o, d, auto_reset);
                    history_append_lane(h, K, i, n, pob, a, o, d, auto_reset);
                    pob = (d && auto_reset) ? fresh_ob : o;
                    if (!auto_reset) was_done = d;
                }
                action[row] = a; ob[row] = o; done[row] = (uint8_t)d;
                if (rk == OR_REWARD_I32) ((int32_t *)reward)[row] = (int32_t)r;
                else ((float *)reward)[row] = (float)r;
            }
            or_env_pack(&e, w);
            for (int j = 0; j < W; j++) state[(int64_t)j * n + i] = w[j];
            prev_ob[i] = pob;
        }
    }
}
