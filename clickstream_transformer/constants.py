"""Drop-in alias of the reference module clickstream_transformer/constants.py -> MI355X implementation."""
from bert4clickpath_amd.clickstream_transformer.constants import *          # noqa: F401,F403
from bert4clickpath_amd.clickstream_transformer import constants as _impl  # noqa: F401
