"""Live cross-check of the oracle against the UNMODIFIED reference, on fresh seeds every run.
Only runs where /root/reference exists (the build container); skipped on the GPU box.  The committed
fixtures in tests/golden/ are the portable form of the same check."""
import os

import numpy as np
import pytest

from conftest import REPO

REF = os.environ.get("GYM_POMDP_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "gym_pomdp")), reason="reference not present")

CASES = [("rock", {}), ("rock", dict(board_size=11, num_rocks=11)), ("stochrock", {}), ("tag", {}),
         ("tag", dict(num_opponents=3)), ("tag", dict(move_prob=.3)), ("tag", dict(num_opponents=2, move_prob=.55)), ("battleship", {}), ("battleship", dict(board_size=(7, 9), max_len=4)),
         ("tiger", {}), ("network", {}), ("network", dict(n_machines=13, problem_type=3)),
         ("network", dict(n_machines=7, problem_type=2))]


@pytest.fixture(scope="module")
def harness():
    import sys
    sys.dont_write_bytecode = True
    from oracle.ref_harness import harness as h
    h.load_reference()
    return h


@pytest.mark.parametrize("env,kw", CASES, ids=["%s-%d" % (c[0], i) for i, c in enumerate(CASES)])
@pytest.mark.filterwarnings("ignore:invalid value encountered:RuntimeWarning")   # the reference's own 0/0 in its side statistics (rock.py:191, 501)
def test_fresh_seed_mode_a(oracle_lib, harness, env, kw):
    """np.random.seed(s) trace of the reference == oracle on an emulated MT19937, seeds drawn at run time."""
    o = oracle_lib.OracleEnv(env, **kw)
    rs = np.random.RandomState(int.from_bytes(os.urandom(4), "little"))
    for _ in range(2):
        seed = int(rs.randint(0, 2 ** 31))
        acts = rs.randint(o.n_actions, size=400)
        if env == "tag":
            acts = np.where(rs.uniform(size=400) < 0.35, 4, acts)
        ref = harness.trace_mode_a(env, kw, seed, acts)
        got = o.trace_mt(seed, acts)
        for k in ("ob", "reward", "done", "state_pre", "state", "reset_ob"):
            assert np.array_equal(np.asarray(ref[k]), got[k].astype(np.asarray(ref[k]).dtype)), (env, seed, k)


@pytest.mark.parametrize("env,kw", CASES, ids=["%s-%d" % (c[0], i) for i, c in enumerate(CASES)])
@pytest.mark.filterwarnings("ignore:invalid value encountered:RuntimeWarning")
def test_fresh_seed_mode_b(oracle_lib, harness, env, kw):
    """Philox-injected trace of the reference == oracle batch drivers, seed / lanes / t drawn at run time."""
    o = oracle_lib.OracleEnv(env, **kw)
    rs = np.random.RandomState(int.from_bytes(os.urandom(4), "little"))
    seed = int(rs.randint(0, 2 ** 62))
    lane0 = int(rs.randint(0, 2 ** 31))
    t0 = int(rs.randint(0, 2 ** 40))
    L, T = 12, 40
    acts = rs.randint(o.n_actions, size=(L, T))
    ref = harness.trace_mode_b(env, kw, seed, range(lane0, lane0 + L), acts, t0=t0)
    st = o.new_state(L)
    assert np.array_equal(o.batch_reset(st, seed, lane0, t0), ref["ob0"])
    for i in range(T):
        ob, rew, done, _ = o.batch_step(st, acts[:, i], seed, lane0, t0 + 1 + i)
        assert np.array_equal(ob, ref["ob"][:, i]) and np.array_equal(done, ref["done"][:, i])
        assert np.array_equal(rew, ref["reward"][:, i].astype(o.reward_dtype))
        comp = np.array(ref["state"][:, i])
        if env == "tag":
            comp[:, -1] = np.maximum(comp[:, -1], -64)
        assert np.array_equal(o.batch_compact(st), comp), (env, i)
