"""One-off: the model-level GPU tests with every optional fusion switched off (legacy kernels stay correct)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest
from bert4clickpath_amd import ops
ops.flash_ce = False
ops.fused_ln = False
ops.sorted_embed_bwd = False
ops.tn_deterministic = False
sys.exit(pytest.main(['-q', '-x', '-m', 'gpu', 'tests/test_gpu_model.py', 'tests/test_gpu_fullsize.py', 'tests/test_gpu_parallel.py',
                      '-k', 'not encoder_stack and not sorted']))
