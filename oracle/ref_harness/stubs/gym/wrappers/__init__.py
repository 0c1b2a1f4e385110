from gym.wrappers.order_enforcing import OrderEnforcing  # noqa: F401
