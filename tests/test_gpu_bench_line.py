"""bench.py's one JSON line: run it as the driver does (a child process, --gpus 1) on a reduced workload and check that
every key of the contract is there, that the numbers agree with each other, and that the library in the tree was the one
loaded.  The workload is cut down (512 sequences, 2 layers, V = 8,192) so the case costs seconds; the full-size line is the
driver's BENCH file and profiles/r02_bench_*.json."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

TOP = ['metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
       'dtype', 'data', 'config', 'roofline', 'cpu_baseline']
ROOF = ['bound', 'achieved', 'peak', 'unit', 'frac', 'traffic']
CPU = ['value', 'unit', 'cores', 'kind', 'sample']


def _run(extra):
    env = dict(os.environ)
    env.pop('RANK', None), env.pop('WORLD_SIZE', None), env.pop('LOCAL_RANK', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '2', '--batch', '512',
           '--vocab', '8192', '--layers', '2', '--n_batches', '2', '--eval_steps', '2', '--full_steps', '2', '--record_steps', '1'] + extra
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, 'bench.py must print exactly one JSON line, got %d' % len(lines)
    return json.loads(lines[0])


def test_bench_line_has_the_contract_keys_and_consistent_numbers():
    line = _run(['--cpu_rows', '16', '--cpu_seconds', '2'])
    for k in TOP:
        assert k in line, k
    assert line['metric'].startswith('masked-items/sec') and line['unit'] == 'masked-items/s'
    assert line['n_gpus'] == 1 and line['steps'] == 4 and line['warmup'] == 2
    assert line['higher_is_better'] is True and line['scaling'] == 'weak' and line['vs_baseline'] is None
    assert line['dtype'] == 'bf16' and 'synthetic' in line['data']
    assert 'workload' in line['config'] and 'model' not in line['config']
    assert line['value'] > 0 and line['ms_per_step'] > 0
    # value = masked items of the timed steps / their wall time: the two printed numbers have to agree on the items per step
    items_per_step = line['value'] * line['ms_per_step'] / 1e3
    assert 0.5 * 512 < items_per_step < 512 * 11, items_per_step
    roof = line['roofline']
    for k in ROOF:
        assert k in roof, k
    assert roof['bound'] in ('hbm', 'mfma') and roof['unit'] in ('GB/s', 'TFLOP/s')
    assert roof['achieved'] > 0 and roof['peak'] > 0
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-3 * max(roof['frac'], 1e-6) + 1e-6
    assert 0 < roof['frac'] < 1.0
    cpu = line['cpu_baseline']
    for k in CPU:
        assert k in cpu, k
    assert cpu['kind'] == 'port' and cpu['cores'] >= 1 and cpu['value'] > 0 and cpu['unit'] == line['unit']
    assert line['value'] > cpu['value']


def test_bench_line_without_cpu_leg_still_parses():
    line = _run(['--no_cpu_baseline'])
    assert line['value'] > 0 and 'roofline' in line


@pytest.mark.parametrize('n', [2, 4])
def test_bench_starts_its_own_ranks(n):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: bench.py starts the N ranks itself (a parent that
    has made no GPU call; torch.distributed.run on 127.0.0.1) and prints ONE line with n_gpus = N whose value counts every
    rank's items.  On a one-GPU box the ranks share the card and the exchange goes over gloo; the line says so.  N = 4 is the
    largest rehearsal a one-GPU box of this pool takes (at most 6 processes on the card, this one included); the reducer itself
    is rehearsed at world size 8 on CPU tensors (tests/test_parallel_cpu.py)."""
    import torch
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '4', '--warmup', '2', '--batch', '256',
           '--vocab', '8192', '--layers', '2', '--n_batches', '2', '--record_steps', '1']
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, 'one JSON line from rank 0, got %d' % len(lines)
    line = json.loads(lines[0])
    assert line['n_gpus'] == n and line['config']['global_batch'] == 256 * n and line['config']['parallelism'].startswith('dp%d' % n)
    assert ('REHEARSAL' in line['config']['parallelism']) == (torch.cuda.device_count() < n)
    items_per_step = line['value'] * line['ms_per_step'] / 1e3
    assert 0.5 * 256 * n < items_per_step < 256 * n * 11, items_per_step        # every rank's masked items
    assert 'eval' not in line and 'cpu_baseline' not in line            # N = 1 legs only
