from .clickstream_transformer import ClickstreamTransformer, TransformerInputPrep   # noqa: F401
from .head import (BinaryClassificationHead, ClozeMaskedItemPrediction, ClozeScores,  # noqa: F401
                   MultiLabel_MultiClass_classification, SampledSoftmaxHead, SoftMaxHead)
from .losses import MaskedLoss, binary_crossentropy, sparse_categorical_crossentropy  # noqa: F401
from .metrics import F1Score, MaskedMetric, PositiveRate, PredictedPositives          # noqa: F401
from .transformer import (Encoder, EncoderLayer, MultiHeadAttention, Transformer,   # noqa: F401
                          create_padding_mask, point_wise_feed_forward_network, positional_encoding,
                          scaled_dot_product_attention)
