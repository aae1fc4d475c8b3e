"""Reading the checkpoint packages of the reference's driver.save_model (egs/wsj0-mix/common/src/driver.py:208-226) without handing
the file the full unpickler by default."""
import os

import torch

def load_checkpoint(path, trust_pickle=None):
    """torch.load of a package written by the reference's driver.save_model (config values + `state_dict` [+ optimizer / history]):
    plain containers, numbers, strings and tensors, which the safe unpickler (weights_only=True) reads.  Only a file that needs arbitrary
    pickled objects falls back to the full unpickler, and only when the caller vouches for it -- trust_pickle=True or
    SEPK_TRUST_CHECKPOINTS=1 (checkpoints fetched from the network should not be given that)."""
    import pickle
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except (pickle.UnpicklingError, RuntimeError) as e:
        if trust_pickle is None:
            trust_pickle = os.environ.get("SEPK_TRUST_CHECKPOINTS", "0") == "1"
        if not trust_pickle:
            raise RuntimeError("{}: not loadable with the safe unpickler ({}); pass trust_pickle=True / set SEPK_TRUST_CHECKPOINTS=1 if the "
                               "file comes from a source you trust".format(path, str(e).splitlines()[0])) from e
        return torch.load(path, map_location="cpu", weights_only=False)
