"""Drop-in import surface for the reference's ``model/test_DCNet_model.py`` (inference model):
``model(image, word_id, word_mask, n_frame)`` (test_DCNet.py:373).  The class is the same
``grounding_model``; calling it with a 4th argument selects the centre-frame semantics.  The
inference model has no ``feature_map`` parameters: load its checkpoints with strict=False or
through ``load_state_dict_compat``."""
import random  # noqa: F401

import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn as nn  # noqa: F401
import torch.nn.functional as F  # noqa: F401
from torch.autograd import Variable  # noqa: F401

from dcnet_amd.darknet import *  # noqa: F401,F403
from dcnet_amd.model import (ConvBatchNormReLU, PhraseAttention, RNNEncoder, generate_coord)  # noqa: F401
from dcnet_amd.model import grounding_model as _base


class grounding_model(_base):
    def forward(self, image, word_id, word_mask=None, n_frame=5):
        return super().forward(image, word_id, word_mask, n_frame)
