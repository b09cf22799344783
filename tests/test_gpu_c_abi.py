"""The drop-in boundary from a language that is not Python: examples/c_abi/c_abi_smoke.c is plain C (gcc, C11), includes
include/b4c.h, links libb4c_hip.so and the HIP runtime, and checks the [MASK]-position index generation (bit for bit) and
residual + LayerNorm (fp32, 1e-5) against loops of its own.  No torch anywhere in that process."""
import os
import shutil
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc on this box')
    exe = str(tmp_path / 'c_abi_smoke')
    libdir = os.path.join(ROOT, 'bert4clickpath_amd')
    cmd = [gcc, '-std=c11', '-O2', '-D__HIP_PLATFORM_AMD__', os.path.join(ROOT, 'examples', 'c_abi', 'c_abi_smoke.c'),
           '-I' + os.path.join(ROOT, 'include'), '-I/opt/rocm/include', '-L' + libdir, '-lb4c_hip', '-L/opt/rocm/lib', '-lamdhip64', '-lm',
           '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib', '-o', exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_program_compiles_against_the_header(tmp_path):
    """(no GPU needed) the header is C, every call in the example type-checks against it, the library resolves the symbols"""
    if not os.path.exists(os.path.join(ROOT, 'bert4clickpath_amd', 'libb4c_hip.so')):
        pytest.skip('libb4c_hip.so not built')
    exe = _build(tmp_path)
    out = subprocess.run(['ldd', exe], capture_output=True, text=True).stdout
    assert 'libb4c_hip.so' in out and 'not found' not in out.split('libb4c_hip.so')[1].split('\n')[0]


@pytest.mark.gpu
def test_c_program_runs_the_library_without_python(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert 'mask_positions' in r.stdout and 'bit-exact' in r.stdout and 'largest deviation' in r.stdout
    assert 'rejected call says' in r.stdout and 'multiple of 8' in r.stdout
