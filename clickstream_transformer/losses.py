"""Drop-in alias of the reference module clickstream_transformer/losses.py -> MI355X implementation."""
from bert4clickpath_amd.clickstream_transformer.losses import *          # noqa: F401,F403
from bert4clickpath_amd.clickstream_transformer import losses as _impl  # noqa: F401
