"""Drop-in import surface for the reference's ``model/darknet.py``."""
from dcnet_amd.darknet import *  # noqa: F401,F403
from dcnet_amd.darknet import Darknet, parse_model_config, create_modules  # noqa: F401
from dcnet_amd.model import ConvBatchNormReLU  # noqa: F401
