"""Drop-in import surface for the reference's ``model/DCNet_model.py``:
``from model.DCNet_model import *`` (train_DCNet.py:40) gives ``grounding_model`` backed by the
MI355X kernel library, plus the names the reference leaks through its own star-imports."""
import random  # noqa: F401  (train_DCNet.py relies on these being re-exported)

import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn as nn  # noqa: F401
import torch.nn.functional as F  # noqa: F401
from torch.autograd import Variable  # noqa: F401

from dcnet_amd.darknet import *  # noqa: F401,F403
from dcnet_amd.model import (ConvBatchNormReLU, PhraseAttention, RNNEncoder, generate_coord,  # noqa: F401
                             grounding_model)
