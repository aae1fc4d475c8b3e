import warnings

from modules.norm import *  # noqa: F401,F403

warnings.warn("Use modules.norm instead.", FutureWarning)
