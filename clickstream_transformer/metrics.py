"""Drop-in alias of the reference module clickstream_transformer/metrics.py -> MI355X implementation."""
from bert4clickpath_amd.clickstream_transformer.metrics import *          # noqa: F401,F403
from bert4clickpath_amd.clickstream_transformer import metrics as _impl  # noqa: F401
