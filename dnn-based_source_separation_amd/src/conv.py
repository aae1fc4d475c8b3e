"""Legacy import location of the convolution modules (`import conv`); everything lives in `modules.conv`."""
import warnings as _warnings

import modules.conv as _new_home

globals().update({_n: getattr(_new_home, _n) for _n in dir(_new_home) if not _n.startswith("_")})
_warnings.warn("`conv` is a legacy alias: import `modules.conv`", FutureWarning, stacklevel=2)
