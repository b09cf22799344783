"""Drop-in alias of the reference module clickstream_transformer/training_utils.py -> MI355X implementation."""
from bert4clickpath_amd.clickstream_transformer.training_utils import *          # noqa: F401,F403
from bert4clickpath_amd.clickstream_transformer import training_utils as _impl  # noqa: F401
