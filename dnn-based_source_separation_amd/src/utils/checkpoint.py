"""Reading the checkpoint packages of the reference's driver.save_model (egs/wsj0-mix/common/src/driver.py:208-226) without handing
the file the full unpickler by default."""
import os

import torch

def _numpy_scalar_globals():
    """what a reference-written package may hold besides tensors: numpy scalars (`best_loss` as np.float64 ...) -- data, not code"""
    try:
        import numpy as np
        out = [np.dtype, np.float64, np.float32, np.int64, np.int32, np.bool_]
        core = getattr(np, "_core", None) or getattr(np, "core")
        out.append(core.multiarray.scalar)
        out += [type(np.dtype(t)) for t in ("float64", "float32", "int64", "int32", "bool")]
        return out
    except Exception:                            # noqa: BLE001 -- numpy layout differences only cost the convenience
        return []


def load_checkpoint(path, trust_pickle=None):
    """torch.load of a package written by the reference's driver.save_model (config values + `state_dict` [+ optimizer / history]):
    plain containers, numbers, strings and tensors, which the safe unpickler (weights_only=True) reads.  Only a file that needs arbitrary
    pickled objects falls back to the full unpickler, and only when the caller vouches for it -- trust_pickle=True or
    SEPK_TRUST_CHECKPOINTS=1 (checkpoints fetched from the network should not be given that)."""
    import pickle
    try:
        with torch.serialization.safe_globals(_numpy_scalar_globals()):
            return torch.load(path, map_location="cpu", weights_only=True)
    except (EOFError, OSError):
        raise                                   # truncated / unreadable file: that is what it is, not a trust question
    except (pickle.UnpicklingError, RuntimeError) as e:
        if "PytorchStreamReader" in str(e) or "failed finding central directory" in str(e):
            raise RuntimeError("{}: corrupt or truncated checkpoint ({})".format(path, str(e).splitlines()[0])) from e
        if trust_pickle is None:
            trust_pickle = os.environ.get("SEPK_TRUST_CHECKPOINTS", "0") == "1"
        if not trust_pickle:
            raise RuntimeError("{}: not loadable with the safe unpickler ({}); pass trust_pickle=True / set SEPK_TRUST_CHECKPOINTS=1 if the "
                               "file comes from a source you trust".format(path, str(e).splitlines()[0])) from e
        return torch.load(path, map_location="cpu", weights_only=False)
