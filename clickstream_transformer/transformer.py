"""Drop-in alias of the reference module clickstream_transformer/transformer.py -> MI355X implementation."""
from bert4clickpath_amd.clickstream_transformer.transformer import *          # noqa: F401,F403
from bert4clickpath_amd.clickstream_transformer import transformer as _impl  # noqa: F401
