"""Re-run one drawn case of tests/test_gpu_properties.py::test_whole_model_random_configurations_fp32 (hypothesis' inner test) with
given arguments.  usage: python scratch/prop_repro.py seed B S L H dh V trunk two packed dff"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from tests import test_gpu_properties as T
from bert4clickpath_amd import ops
a = sys.argv[1:]
kw = dict(seed=int(a[0]), B=int(a[1]), S=int(a[2]), L=int(a[3]), H=int(a[4]), dh=int(a[5]), V=int(a[6]), trunk=eval(a[7]), two=a[8],
          packed=a[9] == 'True', dff=int(a[10]))
print(kw)
T.test_whole_model_random_configurations_fp32.hypothesis.inner_test(ops=ops, **kw)
print('passed')
