from .util import *  # noqa: F401,F403
