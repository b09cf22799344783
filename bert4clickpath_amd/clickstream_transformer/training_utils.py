"""load_vocabulary (reference: clickstream_transformer/training_utils.py:5-12).  The LR schedules and
Keras callbacks of that file are outside the hot path."""
import os


def load_vocabulary(vocab_file):
    """Lines of the vocabulary file, stripped.  A directory raises IsADirectoryError as the reference does."""
    if os.path.isdir(vocab_file):
        raise IsADirectoryError('%s is a directory.' % vocab_file)
    with open(vocab_file, 'r') as f:
        return [ln.strip() for ln in f.readlines()]
