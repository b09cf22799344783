"""Drop-in alias of the reference module clickstream_transformer/head.py -> MI355X implementation."""
from bert4clickpath_amd.clickstream_transformer.head import *          # noqa: F401,F403
from bert4clickpath_amd.clickstream_transformer import head as _impl  # noqa: F401
