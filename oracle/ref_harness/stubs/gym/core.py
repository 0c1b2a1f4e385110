class Env(object):
    """Old-gym duck type: reset() -> ob, step(a) -> (ob, reward, done, info)."""
    metadata = {"render.modes": []}
    reward_range = (-float("inf"), float("inf"))
    action_space = None
    observation_space = None

    def step(self, action):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def render(self, mode="human"):
        raise NotImplementedError

    def close(self):
        return

    def seed(self, seed=None):
        return []
