from .clickstream_transformer import ClickstreamTransformer, TransformerInputPrep   # noqa: F401
from .head import SoftMaxHead                                                       # noqa: F401
from .losses import MaskedLoss, sparse_categorical_crossentropy                     # noqa: F401
from .transformer import (Encoder, EncoderLayer, MultiHeadAttention, Transformer,   # noqa: F401
                          create_padding_mask, point_wise_feed_forward_network, positional_encoding,
                          scaled_dot_product_attention)
