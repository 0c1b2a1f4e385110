import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a minute or more on the GPU box (still part of -m gpu; deselect with -m 'gpu and not slow')")


def golden_manifest():
    with open(os.path.join(GOLDEN, "MANIFEST.json")) as f:
        return json.load(f)


def golden_cases():
    """[(case, env, kwargs)] with tuple-valued kwargs restored."""
    out = []
    for case, env, kw in golden_manifest()["cases"]:
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in kw.items()}
        out.append((case, env, kw))
    return out


def load_golden(mode, case):
    return dict(np.load(os.path.join(GOLDEN, "mode%s_%s.npz" % (mode, case))))


def saturate_tag_compact(env, state):
    """The packed Tag state keeps num_opp in 7 signed bits, saturating at -64
    (every negative value behaves identically)."""
    if env == "tag":
        state = np.array(state, dtype=np.int64)
        state[..., -1] = np.maximum(state[..., -1], -64)
    return state


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle_lib as ol
    ol.build()
    return ol


# ---- heuristic policy support: shared by the oracle and the GPU parity tests ----------------
def heuristic_replay(case, env, o, ops):
    """Replay a heur_<case>.npz fixture through `ops` (the oracle here, the HIP path in test_gpu_parity.py) and compare
    everything the reference recorded.  `ops` supplies: reset(n, seed, lane0, t) -> ob; preferred() -> (lists, lens);
    legal() -> (lists, lens); target(); pick(lists, lens, t); step(actions, t) -> (ob, reward, done);
    track(prev_ob, actions, ob, done); compact(); belief() -> dict of [K, n] arrays."""
    g = np.load(os.path.join(GOLDEN, "heur_%s.npz" % case))
    seed, t0, lanes = int(g["seed"]), int(g["t0"]), g["lanes"]
    ops.max_size = int(g["max_size"]) if "max_size" in g.files else None     # History(max_size), rock.py:533-544
    L, T = g["action"].shape
    is_rock = env in ("rock", "stochrock")
    starts = [0] + [i for i in range(1, L) if lanes[i] != lanes[i - 1] + 1] + [L]
    for s, e in zip(starts[:-1], starts[1:]):
        n, lane0 = e - s, int(lanes[s])
        pol = (lanes[s:e] & 3)
        prev_ob = ops.reset(n, seed, lane0, t0)
        assert np.array_equal(ops.compact(), saturate_tag_compact(env, g["state0"][s:e]))
        for i in range(T):
            t = t0 + 1 + i
            pl, pn = ops.preferred()
            assert np.array_equal(pn, g["pref_len"][s:e, i]), (case, i)
            w = min(pl.shape[1], g["pref"].shape[2])       # the oracle pads to 160, the HIP path to n_actions
            assert np.array_equal(pl[:, :w], g["pref"][s:e, i, :w]) and (g["pref"][s:e, i, w:] == -1).all(), (case, i)
            if is_rock:
                assert np.array_equal(ops.target(), g["target"][s:e, i]), (case, i)
            ll, ln = ops.legal()
            nA = o.n_actions
            a_pref, a_legal = ops.pick(pl, pn, t), ops.pick(ll, ln, t)
            a_all = ops.pick(np.tile(np.arange(nA, dtype=np.int32), (n, 1)), np.full(n, nA, np.int32), t)
            a = np.where(pol < 2, a_pref, np.where(pol == 2, a_legal, a_all)).astype(np.int32)
            assert np.array_equal(a, g["action"][s:e, i]), (case, i)
            ob, rew, done = ops.step(a, t)
            assert np.array_equal(ob, g["ob"][s:e, i]) and np.array_equal(done, g["done"][s:e, i]), (case, i)
            assert np.array_equal(rew.astype(np.float64), g["reward"][s:e, i]), (case, i)
            ops.track(prev_ob, a, ob, done)
            prev_ob = np.where(done != 0, 0 if is_rock else ops.reset_ob(), ob).astype(np.int32)
            assert np.array_equal(ops.compact(), saturate_tag_compact(env, g["state"][s:e, i])), (case, i)
            if is_rock:
                b = ops.belief()
                for k in ("count", "measured"):
                    assert np.array_equal(b[k].T, g[k][s:e, i]), (case, i, k)
                for k in ("lkv", "lkw", "prob_valuable"):          # float64, bit for bit (any NaN == any NaN: 0/0 can occur)
                    got, want = np.ascontiguousarray(b[k].T), np.ascontiguousarray(g[k][s:e, i])
                    same = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))
                    assert same.all(), (case, i, k)



class OracleHeuristicOps(object):
    def __init__(self, ol, o):
        self.ol, self.o = ol, o

    def reset(self, n, seed, lane0, t):
        self.n, self.seed, self.lane0 = n, seed, lane0
        self.st = self.o.new_state(n)
        ob = self.o.batch_reset(self.st, seed, lane0, t)
        self.is_rock = self.o.name in ("rock", "stochrock")
        self.b = self.ol.Belief(self.o, n) if self.is_rock else None
        self.h = self.ol.HistorySums(self.o, n, max_size=getattr(self, "max_size", None))
        return ob

    def preferred(self):
        return self.o.batch_preferred(self.st, self.h, self.b)

    def legal(self):
        return self.o.batch_legal(self.st)

    def target(self):
        return self.b.select_target(self.st)

    def pick(self, lists, lens, t):
        return self.ol.pick(lists, lens, self.seed, self.lane0, t)

    def step(self, a, t):
        self._pre = self.st.copy()
        ob, rew, done, bad = self.o.batch_step(self.st, a, self.seed, self.lane0, t)
        assert bad == 0
        return ob, rew, done

    def reset_ob(self):
        """observation a Tag lane's reset() returned inside the last auto-resetting step"""
        comp = self.o.batch_compact(self.st)
        return np.where((comp[:, 1:-1] == comp[:, :1]).any(axis=1), 29, comp[:, 0]).astype(np.int32)

    def track(self, prev_ob, a, ob, done):
        if self.b is not None:
            self.b.update(self.st, a, ob, done)
        self.h.append(prev_ob, a, ob, done)

    def compact(self):
        return self.o.batch_compact(self.st)

    def belief(self):
        return {k: getattr(self.b, k) for k, _ in self.ol.Belief.FIELDS}
