import warnings

from modules.conv import *  # noqa: F401,F403

warnings.warn("Use modules.conv instead.", FutureWarning)
