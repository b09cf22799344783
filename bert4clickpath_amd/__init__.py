"""bert4clickpath_amd: MI355X-native (gfx950) BERT4Rec forward / Cloze-training path of
MiladShahidi/BERT4ClickPath behind the reference's own class API.  Hot path = hand-written HIP
kernels in libb4c_hip.so (include/b4c.h); this package is the host side."""
from ._lib import B4CError, LIB_PATH, lib          # noqa: F401

__version__ = '0.1.0'
