"""Drop-in alias of the reference module clickstream_transformer/clickstream_transformer.py -> MI355X implementation."""
from bert4clickpath_amd.clickstream_transformer.clickstream_transformer import *          # noqa: F401,F403
from bert4clickpath_amd.clickstream_transformer import clickstream_transformer as _impl  # noqa: F401
