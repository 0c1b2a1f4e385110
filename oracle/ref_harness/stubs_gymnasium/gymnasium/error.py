class Error(Exception):
    pass


class UnregisteredEnv(Error):
    pass


class NameNotFound(UnregisteredEnv):
    pass


class ResetNeeded(Error):
    pass
