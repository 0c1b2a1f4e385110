// envs/tag.hip.h — Tag (gym_pomdp/envs/tag.py): the lane functions the generic kernels of step_impl.hip.h / fused_impl.hip.h / planner.hip call.
// Included by envs.hip.h (which holds the Env interface description and the shared helpers).
#pragma once
#include "../envs_common.hip.h"

namespace pomdp {

struct TagEnv {
    using Params = pomdp_tag_params;
    using Reward = float;
    static constexpr int WORDS = 1;
    static constexpr const char *NAME = "TagEnv";
    static constexpr bool POOLED_LPT2 = true;     // kernels_common.hip.h: Finisher<TagEnv, 2, .>
    static constexpr bool QUAD_STEP = false;   // step_impl.hip.h: step_quad_kernel
    static constexpr bool HAS_ROCKS = false;   // a bounded History keeps a window of transitions for this env (history_push)
    static constexpr bool POOLED_ANY_LPT = false;
    static constexpr bool QUAD_SENSOR = false;
    // The T-shaped board never changes (tag.py:36-78): two small LDS tables replace the coordinate arithmetic of the
    // hot step — cell -> x | y << 4, and (cell, move N0 E1 S2 W3) -> the cell the move leads to, or the cell itself
    // when that square does not exist.  Every workgroup computes them once (threads 0-127, one entry each).
    struct Shared { uint8_t xy[32]; uint8_t mv[32 * 4]; };
    struct State { uint32_t w; };

    static __device__ __forceinline__ int n_actions(const Params &) { return 5; }
    // the reward byte of a Packed trajectory record (traj_out.hip.h): the reward itself (-1, -10 or +10: tag.py:112-136), an int8
    static __device__ __forceinline__ uint32_t reward_code(Reward r) { return (uint32_t)(int)r; }
    // ... and back, as the float64 the reference's callers add up (traj_out.hip.h: the Returns sink)
    static __device__ __forceinline__ double code_reward(uint32_t code) { return (double)(int8_t)code; }
    static __device__ __forceinline__ void load(State &st, const uint32_t *state, int64_t, uint32_t i) { st.w = ld_stream(state + i); }
    static __device__ __forceinline__ void store(const State &st, uint32_t *state, int64_t, uint32_t i, bool) { st_stream(state + i, st.w); }

    // tag.py:52-57 get_tag_coord, 59-66 get_index, 46-50 is_inside
    static __device__ __forceinline__ void coord(int idx, int &x, int &y)
    {
        if (idx < 20) { x = idx % 10; y = idx / 10; }
        else { idx -= 20; x = idx % 3 + 5; y = idx / 3 + 2; }
    }
    static __device__ __forceinline__ int index(int x, int y) { return y < 2 ? y * 10 + x : 20 + (y - 2) * 3 + x - 5; }
    static __device__ __forceinline__ bool inside(int x, int y)
    {
        return y >= 2 ? (x >= 5 && x < 8 && y < 5) : (x >= 0 && x < 10 && y >= 0);
    }
    static __device__ __forceinline__ void stage(Shared &sh, const Params &, int tid)
    {
        if (tid < 128) {
            const int cell = min(tid >> 2, 28), d = tid & 3;
            int x, y;
            coord(cell, x, y);
            const int nx = x + (d == 1) - (d == 3), ny = y + (d == 0) - (d == 2);
            sh.mv[tid] = (uint8_t)(inside(nx, ny) ? index(nx, ny) : cell);
            if (d == 0) sh.xy[tid >> 2] = (uint8_t)(x | (y << 4));
        }
    }
    static __device__ __forceinline__ int num_opp(uint32_t w) { return (int)w >> 25; } // sign-extending
    static __device__ __forceinline__ uint32_t with_num_opp(uint32_t w, int no)
    {
        no = no < -64 ? -64 : no;
        return (w & 0x01FFFFFFu) | ((uint32_t)no << 25);
    }
    // tag.py:219-226
    static __device__ __forceinline__ int sample_ob(const Params &p, uint32_t w, int a)
    {
        const uint32_t agent = w & 31u;
        int ob = (int)agent;
        if (a < 4)
            for (int j = 0; j < p.num_opponents; ++j)
                if (((w >> (5 + 5 * j)) & 31u) == agent) ob = p.obs_cells;
        return ob;
    }

    // tag.py:97-102 reset, 181-193 _get_init_state, 43-44 sample = randint(0, 29)
    static __device__ __forceinline__ int reset(const Shared &, const Params &p, State &st, const RngKey &key,
                                                uint32_t lane)
    {
        WordStream ws(key, lane, POMDP_STREAM_RESET);
        uint32_t w = ws.randint(29u);
        for (int j = 0; j < p.num_opponents; ++j) w |= ws.randint(29u) << (5 + 5 * j);
        st.w = with_num_opp(w, p.num_opponents);
        return sample_ob(p, st.w, 0);
    }
    static __device__ __forceinline__ int reset_ob(const Params &p, const State &st) { return sample_ob(p, st.w, 0); }

    // ---- word contract of the one-opponent game (ABI 13, include/pomdp_hip.h) ------------------------------------------------
    // A step draws only when a TAG fails (the opponent's flight: binomial(1, move_prob), then np.random.choice over 2 or 4
    // moves, tag.py:201-207) or succeeds (the auto-reset that follows: randint(29) per cell, tag.py:181-193) — never both, and
    // a fifth of the lane-steps at most.  Both read ONE word W: the lane's element (lane % 4) of block 0 of the QUAD's STEP
    // stream (counter word 0 = lane / 4).  Flight: the double's high word is W (its low word, read on a tie of the top 27 bits
    // only, the element of block 1); the choice's word is W again — it uses bits 0-1, the double bits 5-31.  Auto-reset:
    // attempt i of the masked-rejection draws reads bits 5 i .. 5 i + 4 of W (i < 6), later attempts the lane's own RESET stream
    // from its first word on (4 x 10^-5 of the resets).  One block serves four lanes; reset() itself (pomdp_tag_reset) and
    // games with more opponents keep the sequential per-lane streams.
    static constexpr int QUAD_WORD = 1;                                        // fused_impl.hip.h / planner.hip: the loops that time-share the block
    static __device__ __forceinline__ uint4 quad_block(const RngKey &key, uint32_t lane, uint32_t block)
    {
        return stream_block(key, lane >> 2, POMDP_STREAM_STEP, block);
    }
    static __device__ __forceinline__ uint32_t elem(const uint4 &b, uint32_t e) { return e == 0 ? b.x : e == 1 ? b.y : e == 2 ? b.z : b.w; }
    static __device__ __forceinline__ void auto_reset_word(const Params &p, State &st, uint32_t W, const RngKey &key, uint32_t lane)
    {
        uint32_t w = 0; int have = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const uint32_t v = (W >> (5 * i)) & 31u;
            if (have < 2 && v <= 28u) { w |= v << (5 * have); ++have; }
        }
        if (have < 2) {                                                        // five of six attempts rejected
            WordStream ws(key, lane, POMDP_STREAM_RESET);
            while (have < 2) {
                const uint32_t v = ws.next32() & 31u;
                if (v <= 28u) { w |= v << (5 * have); ++have; }
            }
        }
        st.w = with_num_opp(w, 1);
    }
    // Called convergently by every lane of the wave; `fresh` marks the lanes that start a new episode inside a step's call.
    static __device__ __forceinline__ void reset_where(const Shared &sh, const Params &p, State &st, bool fresh,
                                                       const RngKey &key, uint32_t lane)
    {
        if (p.num_opponents == 1) {                                            // wave-uniform
            if (__any(fresh)) {
                const uint32_t W = elem(quad_block(key, lane, 0u), lane & 3u);
                if (fresh) auto_reset_word(p, st, W, key, lane);
            }
        } else if (fresh) reset(sh, p, st, key, lane);
    }
    static __device__ __forceinline__ void reset_where_chain(const Shared &sh, const Params &p, State &st, bool fresh,
                                                             const RngKey &key, uint32_t lane, const RngKey &akey,
                                                             uint32_t n_actions, int &next_action)
    {
        reset_where_chain_default<TagEnv>(sh, p, st, fresh, key, lane, akey, n_actions, next_action);
    }

    // tag.py:228-229: every action is legal
    static __device__ __forceinline__ int legal_count(const Shared &, const Params &p, const State &) { return n_actions(p); }
    static __device__ __forceinline__ int legal_nth(const Shared &, const Params &, const State &, int idx) { return idx; }

    // tag.py:231-243 _generate_preferred as a bitmask (ascending order is the reference's list order); tag.py:68-74
    // is_corner, coord.py:75-77 opposite.  history.size == 0 gives the legal list (all five actions).
    static __device__ __forceinline__ uint32_t preferred_mask(const Shared &sh, const Params &p, const State &st,
                                                              const pomdp_rock_belief &, const pomdp_history &h,
                                                              int64_t n, uint32_t i)
    {
        return preferred_mask(sh, p, st, h, n, i, 0u, 0u, ld_stream(h.size + i));
    }
    static __device__ __forceinline__ uint32_t preferred_mask(const Shared &sh, const Params &p, const State &st,
                                                              const pomdp_history &h, int64_t n, uint32_t i, uint32_t ck,
                                                              uint32_t mv, int hsize)
    {
        return preferred_mask(sh, p, st, h, n, i, ck, mv, hsize, hsize ? h.last_action[i] : -1, hsize ? h.last_ob[i] : -1);
    }
    // with history[-1].action / .ob already in registers (the fused multi-step loop)
    static __device__ __forceinline__ uint32_t preferred_mask(const Shared &, const Params &, const State &st,
                                                              const pomdp_history &, int64_t, uint32_t, uint32_t,
                                                              uint32_t, int hsize, int last_action, int last_ob)
    {
        if (hsize == 0) return 0x1Fu;
        const int agent = (int)(st.w & 31u);
        int x, y;
        coord(agent, x, y);
        const bool corner = y < 2 ? (x == 0 || x == 9) : (y == 4 && (x == 5 || x == 7));
        if (last_ob == 29 && corner) return 1u << 4;               // grid.n_tiles, whatever obs_cells was set to
        const int la = last_action;
        uint32_t m = 0;
        const int dx[4] = {0, 1, 0, -1}, dy[4] = {1, 0, -1, 0};
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (la != ((d + 2) & 3) && inside(x + dx[d], y + dy[d])) m |= 1u << d;
        return m;
    }

    // tag.py:209-217 _compute_prob
    static __device__ __forceinline__ double compute_prob(const Shared &, const Params &p, const State &st, int, int ob)
    {
        const uint32_t w = st.w, agent = w & 31u;
        if (ob == p.obs_cells)
            for (int j = 0; j < p.num_opponents; ++j)
                if (((w >> (5 + 5 * j)) & 31u) == agent) return 1.0;
        return ob == (int)agent ? 1.0 : 0.0;
    }

    // tag.py:108-143 with one opponent (the default and the benchmark configuration), branch-free: under a random
    // policy every wave holds both moves and TAGs, so both outcomes are evaluated and selected.  Only a failed TAG
    // on a live opponent draws random numbers: binomial(1, move_prob), then np.random.choice over a list whose length is 2
    // or 4, i.e. randint with an exact mask (one word, no rejection) — from the lane's word of the quad's block (above).  The
    // step is therefore split: `pre` does everything but the opponent's flight and says whether the draw is needed,
    // `flee_word` applies it.
    struct Flight { uint32_t list; int oi; bool need; };
    // tag.py:260-280 `_admissable_actions`: the list the eight appends build depends only on the signs of
    // (opponent - agent) in x and y.  Entry k = 3 (sign dx + 1) + (sign dy + 1), four 2-bit moves each (N0 E1 S2 W3,
    // the reference's order); the two-element lists of the diagonal cases are stored twice over, so that
    // `word & 3` picks from them exactly as randint(2)'s `word & 1` does.  k = 4 (same cell) never draws.
    static constexpr uint64_t ADMISSIBLE_LO = 0x61993100adccecbbull;   // k = 0..7
    static constexpr uint32_t ADMISSIBLE_8 = 0x11u;                     // k = 8
    template <class RT>
    static __device__ __forceinline__ void step_one_opponent_pre(const Shared &sh, const Params &p, State &st, int a,
                                                                 int &ob, RT &rew, int &done, Flight &f)
    {
        const uint32_t w = st.w;
        const int agent = (int)(w & 31u), oi = (int)((w >> 5) & 31u), no = num_opp(w);
        const int axy = sh.xy[agent], oxy = sh.xy[oi];
        // a < 4: the agent moves if the target cell exists (tag.py:112-117)
        const uint32_t agent_m = sh.mv[4 * agent + (a & 3)];
        // a == 4 (tag.py:119-134): tagged iff co-located; otherwise the opponent may flee (tag.py:201-207, 260-280)
        const bool colocated = oi == agent;
        const int dx = (oxy & 15) - (axy & 15), dy = (oxy >> 4) - (axy >> 4);
        const int sx1 = min(max(dx, -1), 1) + 1, sy1 = min(max(dy, -1), 1) + 1;              // v_med3_i32
        const int k = 3 * sx1 + sy1;
        const uint32_t list = k == 8 ? ADMISSIBLE_8 : (uint32_t)(ADMISSIBLE_LO >> (8 * (k & 7))) & 0xFFu;
        const bool tag = a == 4;
        const uint32_t w_tag = with_num_opp(w, no - (int)colocated);
        const uint32_t wn = tag ? w_tag : ((w & ~31u) | agent_m);
        rew = tag ? (colocated ? 10.f : -10.f) : -1.f;
        ob = (!tag && ((wn >> 5) & 31u) == (wn & 31u)) ? p.obs_cells : (int)(wn & 31u);   // tag.py:219-226
        done = num_opp(wn) == 0;
        st.w = wn;
        f.list = list; f.oi = oi;
        f.need = tag && !colocated && no > 0;
    }
    // ---- the same from a (agent cell, opponent cell, action) -> outcome table --------------------------------------------
    // Everything step_one_opponent_pre derives from the two cells and the action — where a move leads, what the agent then
    // sees, whether a TAG finds the opponent, the admissible-move list of its flight — is one 32-bit entry of a table the
    // workgroup builds in LDS when a multi-step launch starts (tag_steps_quad_kernel), indexed [action][opponent << 5 | agent]
    // (the low ten bits of the state word as they are): one lookup instead of three dependent ones plus the sign / list
    // arithmetic.
    //   a <  4: bits 0-4 the agent's cell after the move, bits 14-19 the observation (obs_cells where it now stands on the
    //           opponent, else its cell)
    //   a == 4: bits 5-12 the admissible-move list of the opponent's flight, bit 13 co-located (the TAG succeeds)
    struct StepTab { uint32_t e[5][1024]; };
    static __host__ __device__ __forceinline__ bool tab_ok(const Params &p) { return p.num_opponents == 1 && (unsigned)p.obs_cells < 64u; }
    static __device__ __forceinline__ void build_tab(StepTab &tab, const Shared &sh, const Params &p, int tid)
    {
        for (int slot = tid; slot < 1024; slot += 256) {
            const int agent = min(slot & 31, 28), oi = min(slot >> 5, 28);      // cells >= 29 do not occur: any value will do
            const int axy = sh.xy[agent], oxy = sh.xy[oi];
            const int dx = (oxy & 15) - (axy & 15), dy = (oxy >> 4) - (axy >> 4);
            const int k = 3 * (min(max(dx, -1), 1) + 1) + min(max(dy, -1), 1) + 1;
            const uint32_t list = k == 8 ? ADMISSIBLE_8 : (uint32_t)(ADMISSIBLE_LO >> (8 * (k & 7))) & 0xFFu;
            tab.e[4][slot] = (list << 5) | ((uint32_t)(oi == agent) << 13);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const uint32_t to = sh.mv[4 * agent + a];
                tab.e[a][slot] = to | ((to == (uint32_t)oi ? (uint32_t)p.obs_cells : to) << 14);
            }
        }
    }
    template <class RT>
    static __device__ __forceinline__ void step_one_opponent_tab(const StepTab &tab, State &st, int a, int &ob, RT &rew, int &done,
                                                                 Flight &f)
    {
        const uint32_t w = st.w;
        const uint32_t e = tab.e[a][w & 1023u];
        const int no = num_opp(w);
        const bool tag = a == 4, colocated = (e >> 13) & 1u;
        const uint32_t w_tag = with_num_opp(w, no - (int)colocated);
        const uint32_t wn = tag ? w_tag : ((w & ~31u) | (e & 31u));
        rew = tag ? (colocated ? 10.f : -10.f) : -1.f;
        ob = tag ? (int)(w & 31u) : (int)((e >> 14) & 63u);                   // tag.py:219-226
        done = num_opp(wn) == 0;
        st.w = wn;
        f.list = (e >> 5) & 0xFFu; f.oi = (int)((w >> 5) & 31u);
        f.need = tag && !colocated && no > 0;
    }

    // binomial(1, move_prob) of tag.py:204 from the double's numerator: numpy's inversion gives [U <= thr] for p > .5 and
    // [U > thr] for p <= .5 — the sense is a flag of the params (wave-uniform)
    static __device__ __forceinline__ bool moves(const Params &p, uint64_t k) { return (k <= p.move_thr) != (p.move_gt != 0); }
    // the one-opponent game's flight from the lane's word W of the quad's block (lo(): the double's low word, a tie only)
    template <class LowWord>
    static __device__ __forceinline__ void flee_word(const Shared &sh, const Params &p, State &st, const Flight &f, uint32_t W, LowWord lo)
    {
        const uint32_t pick = (f.list >> (2 * (W & 3u))) & 3u;        // np.random.choice: randint(2 or 4), exact mask
        const uint32_t to = sh.mv[4 * f.oi + (int)pick];               // the cell itself if the square does not exist
        const uint32_t kh = W >> 5, th = (uint32_t)(p.move_thr >> 26);
        bool le = kh < th;                                             // k53 <= move_thr: the top 27 bits decide, but for a tie
        if (f.need && kh == th) le = (lo() >> 6) <= (uint32_t)(p.move_thr & 0x3FFFFFFu);
        if (f.need && (le != (p.move_gt != 0))) st.w = (st.w & ~(31u << 5)) | (to << 5);
    }
    template <class RT>
    static __device__ __forceinline__ void step_one_opponent(const Shared &sh, const Params &p, State &st, int a,
                                                             const RngKey &key, uint32_t lane, int &ob, RT &rew, int &done)
    {
        Flight f;
        step_one_opponent_pre(sh, p, st, a, ob, rew, done, f);
        if (__any(f.need)) {                             // a policy that rarely tags (the heuristic one) rarely pays for the block
            const uint32_t W = elem(quad_block(key, lane, 0u), lane & 3u);
            flee_word(sh, p, st, f, W, [&]() { return elem(quad_block(key, lane, 1u), lane & 3u); });
        }
    }
    // the step and the auto-reset given the lane's word W of the quad's block (the loops that time-share it); games with more
    // opponents (wave-uniform) take the per-lane streams and leave W unread
    template <class RT>
    static __device__ __forceinline__ void step_w(const Shared &sh, const Params &p, State &st, int a, const RngKey &key, uint32_t lane,
                                                  uint32_t W, int &ob, RT &rew, int &done)
    {
        if (p.num_opponents != 1) { step(sh, p, st, a, key, lane, ob, rew, done); return; }
        Flight f;
        step_one_opponent_pre(sh, p, st, a, ob, rew, done, f);
        flee_word(sh, p, st, f, W, [&]() { return elem(quad_block(key, lane, 1u), lane & 3u); });
    }
    static __device__ __forceinline__ void fresh_w(const Shared &sh, const Params &p, State &st, bool fresh, const RngKey &key,
                                                   uint32_t lane, uint32_t W)
    {
        if (!fresh) return;
        if (p.num_opponents == 1) auto_reset_word(p, st, W, key, lane);
        else reset(sh, p, st, key, lane);
    }
    // tag.py:108-143 step, 201-207 move_opponent, 260-280 _admissable_actions
    template <class RT>
    static __device__ __forceinline__ void step(const Shared &sh, const Params &p, State &st, int a,
                                                const RngKey &key, uint32_t lane, int &ob, RT &rew, int &done)
    {
        if (p.num_opponents == 1) { step_one_opponent(sh, p, st, a, key, lane, ob, rew, done); return; }   // wave-uniform
        uint32_t w = st.w;
        const int agent = (int)(w & 31u);
        int ax, ay;
        coord(agent, ax, ay);
        if (a == 4) {
            WordStream ws(key, lane, POMDP_STREAM_STEP);
            bool tagged = false;
            int no = num_opp(w);
            for (int j = 0; j < p.num_opponents; ++j) {
                const int sh = 5 + 5 * j;
                const int oi = (int)((w >> sh) & 31u);
                if (oi == agent) { tagged = true; no -= 1; }
                else if (no > 0) {
                    int ox, oy;
                    coord(oi, ox, oy);
                    // admissible moves, 2 bits each (index into N0 E1 S2 W3), in the reference's list order
                    uint32_t list = 0; int cnt = 0;
                    if (ox >= ax) { list |= 1u << (2 * cnt); ++cnt; }
                    if (oy >= ay) { list |= 0u << (2 * cnt); ++cnt; }
                    if (ox <= ax) { list |= 3u << (2 * cnt); ++cnt; }
                    if (oy <= ay) { list |= 2u << (2 * cnt); ++cnt; }
                    if (ox == ax && oy > ay) { list |= 0u << (2 * cnt); ++cnt; }
                    if (oy == ay && ox > ax) { list |= 1u << (2 * cnt); ++cnt; }
                    if (ox == ax && oy < ay) { list |= 2u << (2 * cnt); ++cnt; }
                    if (oy == ay && ox < ax) { list |= 3u << (2 * cnt); ++cnt; }
                    if (moves(p, ws.next_k53())) {                        // binomial(1, move_prob)
                        const uint32_t pick = (list >> (2 * ws.randint((uint32_t)cnt))) & 3u; // np.random.choice
                        const int nx = ox + (pick == 1u) - (pick == 3u), ny = oy + (pick == 0u) - (pick == 2u);
                        if (inside(nx, ny)) w = (w & ~(31u << sh)) | ((uint32_t)index(nx, ny) << sh);
                    }
                }
            }
            rew = tagged ? 10.f : -10.f;
            w = with_num_opp(w, no);
        } else {
            rew = -1.f;
            const int nx = ax + (a == 1) - (a == 3), ny = ay + (a == 0) - (a == 2);
            if (inside(nx, ny)) w = (w & ~31u) | (uint32_t)index(nx, ny);
        }
        ob = sample_ob(p, w, a);
        done = num_opp(w) == 0;
        st.w = w;
    }
};

} // namespace pomdp
