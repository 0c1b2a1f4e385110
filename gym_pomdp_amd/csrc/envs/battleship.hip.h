// envs/battleship.hip.h — BattleShip (gym_pomdp/envs/battleship.py): the lane functions the generic kernels of step_impl.hip.h / fused_impl.hip.h / planner.hip call.
// Included by envs.hip.h (which holds the Env interface description and the shared helpers).
#pragma once
#include "../envs_common.hip.h"

namespace pomdp {

template <int MW> // mask words: ceil((cells + 6) / 32)
struct BattleShipEnv {
    using Params = pomdp_battleship_params;
    using Reward = int32_t;
    static constexpr int WORDS = 3 * MW;       // occupied, visited (+ remaining), the NEXT episode's occupied mask
    static constexpr bool HAS_NEXT = true;     // step_impl.hip.h, fused_impl.hip.h: the kernels fetch `next` where a lane may need it (load_next)
    static constexpr const char *NAME = MW == 1 ? "BattleShipEnv<1>" : MW == 2 ? "BattleShipEnv<2>" : MW == 3 ? "BattleShipEnv<3>" : "BattleShipEnv<4>";
    static constexpr bool POOLED_LPT2 = false;
    static constexpr bool QUAD_STEP = false;   // step_impl.hip.h: step_quad_kernel
    static constexpr bool HAS_ROCKS = false;   // a bounded History keeps a window of transitions for this env (history_push)
    static constexpr bool POOLED_ANY_LPT = false;
    static constexpr bool QUAD_SENSOR = false;
    struct Shared { int unused; };
    // Each 128-bit mask is two 64-bit registers (never an addressable array or vector: a dynamically indexed
    // one is lowered through LDS by the compiler); bit tests are a 64-bit select and one variable shift.
    struct Mask {
        uint64_t lo, hi;
        __device__ __forceinline__ uint32_t word(int j) const { return (uint32_t)((j < 2 ? lo : hi) >> (32 * (j & 1))); }
        __device__ __forceinline__ void set_word(int j, uint32_t w)
        {
            const uint64_t m = 0xFFFFFFFFull << (32 * (j & 1)), v = (uint64_t)w << (32 * (j & 1));
            if (j < 2) lo = (lo & ~m) | v; else hi = (hi & ~m) | v;
        }
    };
    // Board contract: a lane always holds the board of its NEXT episode as well.  Whenever a board is
    // dealt at call counter t — by reset() (from stream RESET of (lane, t)) or by the auto-reset of a step at t (the cached
    // board moves in) — the following board is drawn from stream NEXT of (lane, t).  An episode's end therefore costs a
    // few selects inside a fused loop, and the rejection loops of all the boards a wave used up during a launch run
    // afterwards, 64 of them side by side (battleship_steps_quad_kernel).  `next` is only loaded where it can be needed.
    struct State { Mask occ, vis, next; };

    static __device__ __forceinline__ void stage(Shared &, const Params &, int) {}
    static __device__ __forceinline__ int n_actions(const Params &p) { return p.x_size * p.y_size; }
    // the reward byte of a Packed trajectory record (traj_out.hip.h): the reward itself (-10 .. cells - 1 <= 121), an int8
    static __device__ __forceinline__ uint32_t reward_code(Reward r) { return (uint32_t)(int)r; }
    // ... and back, as the float64 the reference's callers add up (traj_out.hip.h: the Returns sink)
    static __device__ __forceinline__ double code_reward(uint32_t code) { return (double)(int8_t)code; }
    static __device__ __forceinline__ int reset_ob(const Params &, const State &) { return 0; }   // battleship.py:131-137
    static __device__ __forceinline__ void load(State &st, const uint32_t *state, int64_t n, uint32_t i)
    {
        uint32_t o[4] = {0, 0, 0, 0}, v[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < MW; ++j) { o[j] = ld_stream(state + (int64_t)j * n + i); v[j] = ld_stream(state + (int64_t)(MW + j) * n + i); }
        st.occ.lo = o[0] | ((uint64_t)o[1] << 32); st.occ.hi = o[2] | ((uint64_t)o[3] << 32);
        st.vis.lo = v[0] | ((uint64_t)v[1] << 32); st.vis.hi = v[2] | ((uint64_t)v[3] << 32);
        st.next.lo = st.next.hi = 0;
    }
    static __device__ __forceinline__ void load_next(State &st, const uint32_t *state, int64_t n, uint32_t i)
    {
        uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < MW; ++j) o[j] = ld_stream(state + (int64_t)(2 * MW + j) * n + i);
        st.next.lo = o[0] | ((uint64_t)o[1] << 32); st.next.hi = o[2] | ((uint64_t)o[3] << 32);
    }
    // a step only changes the visited half; the occupied half and the next board are rewritten on reset
    static __device__ __forceinline__ void store(const State &st, uint32_t *state, int64_t n, uint32_t i, bool was_reset)
    {
#pragma unroll
        for (int j = 0; j < MW; ++j) st_stream(state + (int64_t)(MW + j) * n + i, (uint32_t)st.vis.word(j));
        if (was_reset) {
#pragma unroll
            for (int j = 0; j < MW; ++j) {
                st_stream(state + (int64_t)j * n + i, (uint32_t)st.occ.word(j));
                st_stream(state + (int64_t)(2 * MW + j) * n + i, (uint32_t)st.next.word(j));
            }
        }
    }
    // the cached board moves in: a fresh episode (battleship.py:131-137: nothing visited, every ship cell remaining)
    static __device__ __forceinline__ void swap_in(State &st)
    {
        st.occ = st.next;
        st.vis.lo = 0; st.vis.hi = 0;
        st.vis.set_word(MW - 1, (uint32_t)(__popcll(st.occ.lo) + __popcll(st.occ.hi)) << 26);
    }
    static __device__ __forceinline__ bool bit(const Mask &m, int a) { return ((a < 64 ? m.lo : m.hi) >> (a & 63)) & 1ull; }
    static __device__ __forceinline__ void set_bit(Mask &m, int a)
    {
        const uint64_t b = 1ull << (a & 63);
        m.lo |= a < 64 ? b : 0ull;
        m.hi |= a < 64 ? 0ull : b;
    }
    static __device__ __forceinline__ bool occupied(const Params &p, const State &st, int x, int y)
    {
        return (unsigned)x < (unsigned)p.x_size && (unsigned)y < (unsigned)p.y_size && bit(st.occ, y * p.x_size + x);
    }

    // battleship.py:131-137 reset, 167-180 _get_init_state, 195-211 collision, 182-193 mark_ship,
    // coord.py:68-69 Grid.sample, battleship.py:33-37 Ship.__init__ (position word(s) before direction word).
    //
    // The reference's collision() walks L+1 cells from pos and, for each, looks at the cell itself and
    // its N, E, S, W, NE, SE, SW neighbours (Compass[0..7]; NW is never looked at).  Here that is one
    // AND of two 128-bit masks: `blocked` = every cell that has an occupied cell in that 8-neighbourhood
    // (7 shifted copies of the occupancy mask, column-wrap guarded), against the L+1 ship cells; the
    // "pos + dir stays inside for i = 0..L" test reduces to the far end pos + (L+1) dir being inside.
    typedef unsigned __int128 u128;
    // one board from the word stream `stream` of (lane, key's call counter), exactly as the reference's loop consumes it
    static __device__ __forceinline__ u128 board(const Params &p, const RngKey &key, uint32_t lane, uint32_t stream)
    {
        WordStream ws(key, lane, stream);
        const int X = p.x_size, Y = p.y_size;
        u128 col0 = 0;                                     // cells with x == 0
        for (int y = 0; y < Y; ++y) col0 |= (u128)1 << (y * X);
        const u128 colL = col0 << (X - 1);                 // cells with x == X - 1
        u128 occ = 0;
        for (int len = p.max_len; len >= 2; --len) {
            const u128 e = occ & ~col0, w = occ & ~colL;   // sources that may shift one column west / east
            const u128 blocked = occ | (occ >> X) | (occ << X) | (e >> 1) | (w << 1) | (e >> (X + 1)) | (e << (X - 1)) |
                                 (w << (X + 1));
            int a0, dx, dy;
            for (;;) {
                a0 = (int)ws.randint((uint32_t)(X * Y));
                const uint32_t dir = ws.randint(4u);
                dx = (dir == 1u) - (dir == 3u); dy = (dir == 0u) - (dir == 2u);   // Compass N E S W
                const int px = a0 % X, py = a0 / X;
                const int ex = px + (len + 1) * dx, ey = py + (len + 1) * dy;
                if (!((unsigned)ex < (unsigned)X && (unsigned)ey < (unsigned)Y)) continue;
                const int stride = dy * X + dx;                                   // bit distance between ship cells
                const int lo = stride > 0 ? a0 : a0 + len * stride;               // lowest bit of the L+1 checked cells
                const int gap = stride > 0 ? stride : -stride;
                u128 cells = 0;
                for (int i = 0; i <= len; ++i) cells |= (u128)1 << (lo + i * gap);
                if ((cells & blocked) == 0) break;
            }
            const int stride = dy * X + dx;
            for (int i = 0; i < len; ++i) occ |= (u128)1 << (a0 + i * stride);     // mark_ship: L cells from pos
        }
        return occ;
    }
    // reset() at call counter t: this episode's board from stream RESET, the next episode's from stream NEXT
    static __device__ __forceinline__ int reset(const Shared &, const Params &p, State &st, const RngKey &key,
                                                uint32_t lane)
    {
        const u128 nx = board(p, key, lane, POMDP_STREAM_RESET);
        st.next.lo = (uint64_t)nx; st.next.hi = (uint64_t)(nx >> 64);
        swap_in(st);
        const u128 n2 = board(p, key, lane, POMDP_STREAM_NEXT);
        st.next.lo = (uint64_t)n2; st.next.hi = (uint64_t)(n2 >> 64);
        return 0;
    }

    // Wave-cooperative auto-reset (the single-step kernels and the one-lane-per-thread loops): a lane whose episode ended
    // takes its cached board (swap_in) and the wave builds the board after it, from stream NEXT of (lane, t).
    // A BattleShip board is a long sequential rejection loop (about 20 attempts on
    // 10x10, 42 on 5x5) and roughly one wave in five holds a lane that needs one; run per lane it stalls 63
    // other lanes behind ~2000 divergent instructions.  Here the whole wave serves one resetting lane at a
    // time and evaluates up to 64 candidate placements at once:
    //   1. one Philox pass gives a 64-word window of that lane's NEXT stream (lane l holds word c + l);
    //   2. the reference consumes the stream as [position words until one is < n_tiles][direction word],
    //      repeated.  With A = ballot(word is an acceptable position), word l is a *direction* word iff
    //      word l-1 is an accepted position word, i.e. D[l] = A[l-1] & ~D[l-1]: inside every run of ones of
    //      A the roles alternate, which is the "escaped character" recurrence and has a branch-free 64-bit
    //      solution (add-with-carry over the run starts; Langdale & Lemire, "Parsing gigabytes of JSON per
    //      second", §3.1.1).  Lane l is a candidate iff A[l] & ~D[l]; its direction word is lane l+1's;
    //   3. every candidate lane tests its placement with 128-bit mask arithmetic against `blocked`; the
    //      lowest successful lane is the ship the reference would have placed, and the cursor moves just
    //      past its direction word.  No success: the cursor moves past the last fully parsed word.
    // Same words in the same order as board() above, hence the same boards.
    static __device__ __forceinline__ u128 u128_of4(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
    {
        return (u128)(w0 | ((uint64_t)w1 << 32)) | ((u128)(w2 | ((uint64_t)w3 << 32)) << 64);
    }
    static __device__ __forceinline__ u128 u128_of(const uint32_t (&w)[4]) { return u128_of4(w[0], w[1], w[2], w[3]); }
    // The cooperative reset's cell masks in the narrowest type that holds the board (cells + 6 <= 32 MW bits): one
    // 32-bit word for the reference's default 5x5 board, one 64-bit word up to 58 cells, 128 bits beyond — uniform mask
    // arithmetic runs on the scalar unit, where a 128-bit operation is four instructions and a 32-bit one is one.
    using M = typename std::conditional<MW == 1, uint32_t, typename std::conditional<MW == 2, uint64_t, u128>::type>::type;
    static constexpr int MBITS = MW == 1 ? 32 : MW == 2 ? 64 : 128;
    static __device__ __forceinline__ M mask_of(const uint32_t (&w)[4]) { return (M)u128_of(w); }
    // shifts by 1 <= s <= 63 (a board is at most 16 wide): for 128 bits the general form selects between three cases
    static __device__ __forceinline__ M shl_small(M a, int s)
    {
        if constexpr (MW <= 2) return (M)(a << s);
        else {
            const uint64_t lo = (uint64_t)a, hi = (uint64_t)((u128)a >> 64);
            return (M)((u128)(lo << s) | ((u128)((hi << s) | (lo >> (64 - s))) << 64));
        }
    }
    static __device__ __forceinline__ M shr_small(M a, int s)
    {
        if constexpr (MW <= 2) return (M)(a >> s);
        else {
            const uint64_t lo = (uint64_t)a, hi = (uint64_t)((u128)a >> 64);
            return (M)((u128)((lo >> s) | (hi << (64 - s))) | ((u128)(hi >> s) << 64));
        }
    }
    // ---- the same boards, many lanes at once (fused_impl.hip.h: battleship_steps_quad_kernel) -----------------------------
    // board() reshaped for lanes that run it in lockstep: ONE loop in which every lane consumes exactly one word of its
    // stream per iteration — so the Philox blocks are generated under a wave-uniform condition, four iterations per block —
    // and a two-state machine per lane says what the word is: a position word (accepted when <= n_tiles - 1,
    // np.random.randint's masked rejection) or the direction word that follows an accepted position (battleship.py:33-37,
    // 171-176).  The placement test and mark_ship are the mask arithmetic of reset_where(); the column patterns come from
    // `vp` (LDS copy of p.vpat: one 16-byte read per lane whatever its ship length).  Lanes with go == false idle through.
    // Every lane brings its own key (the call counter at which its previous board was dealt).
    struct SeqTables { uint32_t vp[16][4]; };   // rows 0 .. 11 hold p.vpat
    static __device__ __forceinline__ void stage_seq(SeqTables &t, const Params &p, int tid)
    {
        if (tid < 48) t.vp[tid >> 2][tid & 3] = p.vpat[(tid >> 2) % 12][tid & 3];
    }
    static __device__ __forceinline__ M blocked_of(M occ, M col0, M colL, int X)
    {
        // occ and its N, E, S, W, NE, SE, SW shifts (NW excluded): h = {self, E, W}; south side = h << X; north = {self, E} >> X
        const M e1 = (M)((occ & ~col0) >> 1), h = (M)(occ | e1 | (M)((occ & ~colL) << 1));
        return (M)(h | shr_small((M)(occ | e1), X) | shl_small(h, X));
    }
    // the two-state word consumer of one board under construction
    struct Builder {
        M occ, blocked;
        int len, a0;                                                           // len < 2: nothing (left) to place
        bool want_dir;
        __device__ __forceinline__ void start(int max_len) { occ = 0; blocked = 0; len = max_len; a0 = 0; want_dir = false; }
        __device__ __forceinline__ void idle() { occ = 0; blocked = 0; len = 1; a0 = 0; want_dir = false; }
        __device__ __forceinline__ bool busy() const { return len >= 2; }
    };
    struct BuildConsts { int X, Y, cells; uint32_t rmask, inv_x; M col0, colL; };
    static __device__ __forceinline__ BuildConsts build_consts(const Params &p)
    {
        BuildConsts c;
        c.X = p.x_size; c.Y = p.y_size; c.cells = c.X * c.Y;
        __builtin_assume(c.X >= 1 && c.X <= 16 && c.Y >= 1 && c.Y <= 16);
        c.rmask = 0xFFFFFFFFu >> __clz((uint32_t)(c.cells - 1) | 1u);
        c.inv_x = (65536u + (uint32_t)c.X - 1u) / (uint32_t)c.X;               // a / X == (a * inv_x) >> 16 for a < 128, X <= 16
        c.col0 = mask_of(p.col0); c.colL = (M)(c.col0 << (c.X - 1));
        return c;
    }
    // one word of the board's stream: as a direction word it completes the placement from a0, as a position word it is
    // accepted when it is a tile (the lane's state says which of the two it is)
    static __device__ __forceinline__ void feed(Builder &b, const SeqTables &t, const BuildConsts &c, uint32_t w)
    {
        const int X = c.X, len = b.len, a0 = b.a0;
        const bool live = len >= 2;
        const uint32_t dir2 = (w & 3u) << 1;                                   // Compass N E S W: (dx, dy) = (0,1) (1,0) (0,-1) (-1,0),
        const int dx = __builtin_amdgcn_sbfe(0xC4, dir2, 2u), dy = __builtin_amdgcn_sbfe(0x31, dir2, 2u);   // 2-bit signed fields of two constants
        const int py = (int)(((uint32_t)a0 * c.inv_x) >> 16), px = a0 - py * X;
        const int ex = px + (len + 1) * dx, ey = py + (len + 1) * dy, stride = dy * X + dx;
        const bool inside = (unsigned)ex < (unsigned)X && (unsigned)ey < (unsigned)c.Y;
        const int lo = stride > 0 ? a0 : a0 + len * stride;
        const uint32_t *v1 = t.vp[(len + 1) & 15], *v0 = t.vp[len & 15];         // len <= max_len <= 10 (bs_mask_words)
        const uint32_t w1[4] = {v1[0], v1[1], v1[2], v1[3]}, w0[4] = {v0[0], v0[1], v0[2], v0[3]};
        const M vpat1 = mask_of(w1), vpat0 = mask_of(w0);
        const M test = (M)((dx != 0 ? (M)((1ull << (len + 1)) - 1ull) : vpat1) << (lo & (MBITS - 1)));
        const bool place = live && b.want_dir && inside && (test & b.blocked) == 0;
        if (__any(place)) {                                                    // wave-uniform: skip the marking when nobody places
            if (place) {
                const int low = stride > 0 ? a0 : a0 + (len - 1) * stride;
                b.occ |= (M)((dx != 0 ? (M)((1ull << len) - 1ull) : vpat0) << (low & (MBITS - 1)));
                b.len = len - 1;
                b.blocked = blocked_of(b.occ, c.col0, c.colL, X);
            }
        }
        const uint32_t v = w & c.rmask;
        const bool accept = live && !b.want_dir && v <= (uint32_t)(c.cells - 1);
        b.a0 = accept ? (int)v : a0;
        b.want_dir = accept;                                                   // a direction word is always consumed: back to positions
    }

    // Two consecutive words of the board's stream at once — exactly feed(w0) then feed(w1), with ONE placement test: a lane
    // tests a placement at most once per two words whatever its state (waiting for a direction: w0 completes the pending
    // position, then w1 is a position word; waiting for a position: w0 accepted -> w1 is its direction word and completes it,
    // w0 rejected -> w1 is the next position word), and the test — coordinates, bounds, two LDS pattern rows, the mask shift
    // and the `blocked` recomputation behind it — is most of what a word costs (round 4: the builder pool is 60 % of the 5x5
    // board's launch; 4.2 -> 3.x us per step).
    static __device__ __forceinline__ void feed2(Builder &b, const SeqTables &t, const BuildConsts &c, uint32_t w0, uint32_t w1)
    {
        const int X = c.X, len = b.len;
        const bool live = len >= 2, D = b.want_dir;
        const uint32_t v0 = w0 & c.rmask, v1 = w1 & c.rmask;
        const bool acc0 = v0 <= (uint32_t)(c.cells - 1), acc1 = v1 <= (uint32_t)(c.cells - 1);
        const bool test_now = live && (D || acc0);                             // a (position, direction) pair is complete
        const int a0 = D ? b.a0 : (int)v0;
        const uint32_t dir2 = ((D ? w0 : w1) & 3u) << 1;                       // Compass N E S W: (dx, dy) = (0,1) (1,0) (0,-1) (-1,0)
        const int dx = __builtin_amdgcn_sbfe(0xC4, dir2, 2u), dy = __builtin_amdgcn_sbfe(0x31, dir2, 2u);
        const int py = (int)(((uint32_t)a0 * c.inv_x) >> 16), px = a0 - py * X;
        const int ex = px + (len + 1) * dx, ey = py + (len + 1) * dy, stride = dy * X + dx;
        const bool inside = (unsigned)ex < (unsigned)X && (unsigned)ey < (unsigned)c.Y;
        const int lo = stride > 0 ? a0 : a0 + len * stride;
        const uint32_t *v1p = t.vp[(len + 1) & 15], *v0p = t.vp[len & 15];       // len <= max_len <= 10 (bs_mask_words)
        const uint32_t p1[4] = {v1p[0], v1p[1], v1p[2], v1p[3]}, p0[4] = {v0p[0], v0p[1], v0p[2], v0p[3]};
        const M vpat1 = mask_of(p1), vpat0 = mask_of(p0);
        const M test = (M)((dx != 0 ? (M)((1ull << (len + 1)) - 1ull) : vpat1) << (lo & (MBITS - 1)));
        const bool place = test_now && inside && (test & b.blocked) == 0;
        int len_after = len;
        if (__any(place)) {                                                    // wave-uniform: skip the marking when nobody places
            if (place) {
                const int low = stride > 0 ? a0 : a0 + (len - 1) * stride;
                b.occ |= (M)((dx != 0 ? (M)((1ull << len) - 1ull) : vpat0) << (low & (MBITS - 1)));
                len_after = len - 1;
                b.len = len_after;
                b.blocked = blocked_of(b.occ, c.col0, c.colL, X);
            }
        }
        // w1 as a position word: after a completed direction (D), or after a rejected w0
        const bool accept1 = len_after >= 2 && (D || !acc0) && acc1;
        b.a0 = accept1 ? (int)v1 : a0;
        b.want_dir = accept1;
    }

    static __device__ __forceinline__ uint64_t direction_words(uint64_t a)
    {
        const uint64_t even = 0x5555555555555555ull;
        const uint64_t follows = a << 1;                       // words preceded by an acceptable word
        const uint64_t odd_starts = a & ~even & ~follows;      // runs of A that start on an odd bit
        const uint64_t even_start_runs = odd_starts + a;       // carry ripples through those runs
        return (even ^ (even_start_runs << 1)) & follows;
    }
    static __device__ __forceinline__ void reset_where(const Shared &, const Params &p, State &st, bool fresh,
                                                       const RngKey &key, uint32_t lane)
    {
        uint64_t todo = __ballot(fresh);
        if (todo == 0ull) return;                                            // wave-uniform
        if (fresh) swap_in(st);                                              // the cached board; below: the one after it, stream NEXT
        const int me = (int)(threadIdx.x & 63u);
        const int X = p.x_size, Y = p.y_size, cells = X * Y;
        __builtin_assume(X >= 1 && X <= 16 && Y >= 1 && Y <= 16);          // bs_mask_words(): what the launchers let through —
        __builtin_assume(p.max_len >= 2 && p.max_len <= 10);                // 128-bit shifts by X or by a ship length stay below 64
        const uint32_t rmask = 0xFFFFFFFFu >> __clz((uint32_t)(cells - 1) | 1u); // randint(cells) bit-smear mask
        const M col0 = mask_of(p.col0), colL = (M)(col0 << (X - 1));
        const uint32_t inv_x = (65536u + (uint32_t)X - 1u) / (uint32_t)X;   // a / X == (a * inv_x) >> 16 for a < 128, X <= 16
        while (todo != 0ull) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            const uint32_t glane = (uint32_t)__builtin_amdgcn_readlane((int)lane, src);
            int c = 0;                                                        // next unread word of the stream
            // One Philox pass yields 64 consecutive words and a ship consumes only a dozen of them, so the window is
            // kept across ships (and retries): lane l reads word c + l out of the window that starts at c0 through
            // ds_bpermute, as long as at least 32 of its words are still ahead of the cursor
            int c0 = -64;
            uint32_t wword = 0;
            M occ = 0;
            for (int len = p.max_len; len >= 2; --len) {
                __builtin_assume(len <= 10);
                // blocked = occ and its N, E, S, W, NE, SE, SW shifts (NW excluded) in four 128-bit shifts:
                // h = {self, E, W}; south side = h << X (S, SE, SW); north side = {self, E} >> X (N, NE)
                const M e1 = (M)((occ & ~col0) >> 1), h = (M)(occ | e1 | (M)((occ & ~colL) << 1));
                const M blocked = (M)(h | shr_small((M)(occ | e1), X) | shl_small(h, X));
                const M hpat = (M)((1ull << (len + 1)) - 1ull);         // the L+1 checked cells, from bit 0 (L + 1 <= 11)
                const M vpat = mask_of(p.vpat[len + 1]);
                const M vship = mask_of(p.vpat[len]);                      // mark_ship's column pattern: fetched here, not on the success path
                for (;;) {
                    if (c - c0 > 32) {                                        // wave-uniform: refill the window at the cursor
                        const uint32_t wi = (uint32_t)(c + me);
                        const uint4 blk = stream_block(key, glane, POMDP_STREAM_NEXT, (wi >> 2) & 0xFFFFFFu);
                        const uint32_t sel = wi & 3u;
                        wword = sel == 0 ? blk.x : sel == 1 ? blk.y : sel == 2 ? blk.z : blk.w;
                        c0 = c;
                    }
                    const int off = c - c0, nv = 64 - off;                    // nv words of the window lie at or after the cursor
                    const uint32_t word = (uint32_t)__shfl((int)wword, (me + off) & 63, 64);
                    const uint64_t A = __ballot(me < nv && (word & rmask) <= (uint32_t)(cells - 1));
                    const uint64_t D = direction_words(A);
                    const uint64_t cand = A & ~D & ((1ull << (nv - 1)) - 1ull);   // position word with its direction word in the window
                    const uint32_t dirword = (uint32_t)__shfl((int)wword, (me + off + 1) & 63, 64);
                    const int a0 = (int)(word & rmask);
                    const uint32_t dir = dirword & 3u;
                    const int dx = (dir == 1u) - (dir == 3u), dy = (dir == 0u) - (dir == 2u);   // Compass N E S W
                    const int py = (int)(((uint32_t)a0 * inv_x) >> 16), px = a0 - py * X;
                    const int ex = px + (len + 1) * dx, ey = py + (len + 1) * dy;
                    const int stride = dy * X + dx;
                    const bool inside = (unsigned)ex < (unsigned)X && (unsigned)ey < (unsigned)Y;
                    const int lo = stride > 0 ? a0 : a0 + len * stride;
                    const M cellsm = (M)((dx != 0 ? hpat : vpat) << (lo & (MBITS - 1)));
                    const bool ok = ((cand >> me) & 1ull) && inside && (cellsm & blocked) == 0;
                    const uint64_t succ = __ballot(ok);
                    if (succ != 0ull) {
                        const int r = __ffsll((long long)succ) - 1;
                        const int a0w = __builtin_amdgcn_readlane(a0, r), sw = __builtin_amdgcn_readlane(stride, r);
                        // mark_ship: L cells from pos = the L-cell pattern shifted to its lowest cell
                        const int low = sw > 0 ? a0w : a0w + (len - 1) * sw;
                        occ |= (M)(((sw == 1 || sw == -1) ? (M)((1ull << len) - 1ull) : vship) << (low & (MBITS - 1)));
                        c += r + 2;
                        break;
                    }
                    // no placement here: the window's last word is unread only if it is an accepted position word (its
                    // direction word lies in the next window)
                    c += (((A & ~D) >> (nv - 1)) & 1ull) ? nv - 1 : nv;
                }
            }
            if (me == src) { st.next.lo = (uint64_t)occ; st.next.hi = MW > 2 ? (uint64_t)((u128)occ >> 64) : 0ull; }
        }
    }
    static __device__ __forceinline__ void reset_where_chain(const Shared &sh, const Params &p, State &st, bool fresh,
                                                             const RngKey &key, uint32_t lane, const RngKey &akey,
                                                             uint32_t n_actions, int &next_action)
    {
        reset_where_chain_default<BattleShipEnv<MW>>(sh, p, st, fresh, key, lane, akey, n_actions, next_action);
    }

    // battleship.py:157-165 _generate_legal: the unvisited cells, ascending
    static __device__ __forceinline__ uint32_t unvisited(const Params &p, const State &st, int j)
    {
        const int cells = p.x_size * p.y_size, lo = 32 * j;
        const uint32_t valid = cells - lo >= 32 ? 0xFFFFFFFFu : (cells > lo ? (1u << (cells - lo)) - 1u : 0u);
        return ~st.vis.word(j) & valid;
    }
    static __device__ __forceinline__ int legal_count(const Shared &, const Params &p, const State &st)
    {
        int c = 0;
#pragma unroll
        for (int j = 0; j < MW; ++j) c += __popc(unvisited(p, st, j));
        return c;
    }
    static __device__ __forceinline__ int legal_nth(const Shared &, const Params &p, const State &st, int idx)
    {
        int a = 0;
#pragma unroll
        for (int j = 0; j < MW; ++j) {
            uint32_t z = unvisited(p, st, j);
            const int c = __popc(z);
            if (idx >= 0 && idx < c) {
                for (int k = idx; k > 0; --k) z &= z - 1u;
                a = 32 * j + __ffs((int)z) - 1;
            }
            idx -= c;
        }
        return a;
    }

    // no heuristic: _generate_preferred is _generate_legal (0 = fall back to the legal list)
    static __device__ __forceinline__ uint32_t preferred_mask(const Shared &, const Params &, const State &,
                                                              const pomdp_rock_belief &, const pomdp_history &, int64_t,
                                                              uint32_t) { return 0u; }
    static __device__ __forceinline__ uint32_t preferred_mask(const Shared &, const Params &, const State &,
                                                              const pomdp_history &, int64_t, uint32_t, uint32_t, uint32_t,
                                                              int) { return 0u; }
    static __device__ __forceinline__ uint32_t preferred_mask(const Shared &, const Params &, const State &,
                                                              const pomdp_history &, int64_t, uint32_t, uint32_t, uint32_t,
                                                              int, int, int) { return 0u; }
    // battleship.py:80-89 _compute_prob (reads the grid as it is after the shot)
    static __device__ __forceinline__ double compute_prob(const Shared &, const Params &, const State &st, int a, int ob)
    {
        if (ob == 0 && bit(st.vis, a)) return 1.0;
        if (ob == 1 && bit(st.occ, a)) return 1.0;
        return ob == 0 ? 1.0 : 0.0;
    }

    // battleship.py:91-122
    template <class RT>
    static __device__ __forceinline__ void step(const Shared &, const Params &p, State &st, int a,
                                                const RngKey &, uint32_t, int &ob, RT &rew, int &done)
    {
        int remaining = (int)(st.vis.word(MW - 1) >> 26);
        ob = 0; done = 0;
        if (bit(st.vis, a)) rew = -10;
        else {
            rew = -1;
            if (bit(st.occ, a)) { ob = 1; remaining -= 1; }
            set_bit(st.vis, a);
        }
        if (remaining == 0) { rew += p.x_size * p.y_size; done = 1; }
        st.vis.set_word(MW - 1, (st.vis.word(MW - 1) & 0x03FFFFFFu) | ((uint32_t)remaining << 26));
    }
};

} // namespace pomdp
