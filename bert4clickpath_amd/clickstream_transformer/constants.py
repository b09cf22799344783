"""Token / padding constants of the click-stream vocabulary (values follow the reference's
clickstream_transformer/constants.py:1-31; the hot path needs the values, not the code)."""
LABEL_PAD = -1.0
NUM_RESERVED_TOKENS = 10

INPUT_PADDING_TOKEN = '[PAD]'
INPUT_MASKING_TOKEN = '[MASK]'
UNKNOWN_TOKEN = '[UNK]'
CLASSIFICATION_TOKEN = '[CLS]'
SEPARATOR_TOKEN = '[SEP]'
MISSING_EVENT_OR_ITEM_TOKEN = '[NA]'

_NAMED = (INPUT_PADDING_TOKEN, INPUT_MASKING_TOKEN, UNKNOWN_TOKEN, CLASSIFICATION_TOKEN, SEPARATOR_TOKEN,
          MISSING_EVENT_OR_ITEM_TOKEN)
RESERVED_TOKENS = list(_NAMED) + ['[RESERVED_%d]' % i for i in range(len(_NAMED), NUM_RESERVED_TOKENS)]

INPUT_PAD = 0            # id of '[PAD]'
MASK_ID = 1              # id of '[MASK]' (the hot path matches the string '[MASK]' -> id 1)
UNKNOWN_INPUT = 2
CLS = 3
SEP = 4
MISSING_EVENT_OR_ITEM = 5
# the reference defines INPUT_MASK = index('[UNK]') (= 2, constants.py:28); it is unused on the hot path.
INPUT_MASK = UNKNOWN_INPUT

ITEM_EMBEDDING_LAYER_NAME = 'item_embedding_layer'
assert [RESERVED_TOKENS.index(t) for t in _NAMED] == [INPUT_PAD, MASK_ID, UNKNOWN_INPUT, CLS, SEP, MISSING_EVENT_OR_ITEM]
