"""MLP regressors and their circuit-level feature encoders, under the module path the reference uses
(blackwater/library/learning/mlp.py).  Models: :mod:`blackwater.nn.mlp`; encoders: :mod:`.features`."""
from ...nn.mlp import MLP1, MLP2, MLP3  # noqa: F401
from .features import (count_gates_by_rotation_angle, encode_data, encode_data_v2_ecr,  # noqa: F401
                       recursive_dict_loop)


def fix_random_seed(seed=0):
    """Seeds python / numpy / torch (reference: docs/tutorials/mlp.py:112-121)."""
    import os
    import random

    import numpy as np
    import torch

    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
