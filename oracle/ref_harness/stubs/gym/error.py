class Error(Exception):
    pass


class UnregisteredEnv(Error):
    pass


class ResetNeeded(Error):
    pass
