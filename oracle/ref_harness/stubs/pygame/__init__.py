"""`pygame` stand-in: the reference's gui.py evaluates pygame.Color(...) at
class-definition time, which makes pygame an import-time dependency of the env
modules.  Rendering is out of scope; only the import has to succeed."""


def Color(name):
    return name
