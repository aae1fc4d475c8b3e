"""Legacy import location of the normalisation layers (`import norm`); everything lives in `modules.norm`."""
import warnings as _warnings

import modules.norm as _new_home

globals().update({_n: getattr(_new_home, _n) for _n in dir(_new_home) if not _n.startswith("_")})
_warnings.warn("`norm` is a legacy alias: import `modules.norm`", FutureWarning, stacklevel=2)
