"""GPU: the rank body of bench.py as the driver's N > 1 launch runs it -- process group over RCCL, barrier, the
all-gather of the poses, max-over-ranks timing -- rehearsed at world size 1 on the one-GPU box
(MPE_BENCH_FORCE_DIST=1), and the JSON contract of the line it prints.  A world of 2+ GPUs is the driver's to run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _bench(extra_env, *argv):
    env = dict(os.environ)
    env.update(extra_env)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *argv], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_rank_body_under_a_forced_process_group_matches_the_plain_run():
    """Same command with and without the RCCL process group (rank 0 of a world of 1): init_process_group('nccl'),
    the barrier pair around the timed region, all_gather_results and the all-reduce(MAX) of the time all execute on the
    card; the reported workload, step count and parity fields are identical and the throughput is the same to within
    the all-gather's cost."""
    args = ('--gpus', '1', '--steps', '40', '--warmup', '5', '--frames', '256', '--cpu-sample', '4', '--json-steps', '0', '--no-io',
            '--profile-steps', '4')
    plain = _bench({}, *args)
    forced = _bench({'MPE_BENCH_FORCE_DIST': '1', 'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0', 'MASTER_ADDR': '127.0.0.1',
                     'MASTER_PORT': str(_free_port())}, *args)
    for d in (plain, forced):
        assert d['metric'] == plain['metric'] and d['unit'] == 'frames/s' and d['n_gpus'] == 1 and d['steps'] == 40 and d['warmup'] == 5
        assert d['scaling'] == 'weak' and d['higher_is_better'] is True and d['vs_baseline'] is None and d['data'] == 'synthetic'
        assert d['value'] > 0 and abs(d['value'] - 256 / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value']
        assert d['roofline']['bound'] == 'mfma' and 0 < d['roofline']['frac'] < 1
        assert d['cpu_baseline']['kind'] == 'port' and d['cpu_baseline']['value'] > 0
        assert d['parity']['clusters_exact_frac'] == 1.0 and d['parity']['sample_frames'] == 4
    assert plain['config']['backend'] is None and plain['config']['world_size_seen'] == 1
    assert forced['config']['backend'] == 'nccl (RCCL)' and forced['config']['world_size_seen'] == 1
    # the rank body gathers once more after the timed region and compares its own shard inside what it received
    assert plain['config']['gathered_equal_local'] is None and forced['config']['gathered_equal_local'] is True
    # the all-gather of 256 frames of poses is microseconds of a ~1.5 ms step; a forced group that serialised or
    # re-synchronised the step would show as a large drop
    # (measured 0.93-0.97 with the production GEMM; 0.80 with the slow register-staged kernel of the switch matrix)
    assert forced['value'] >= 0.65 * plain['value'], (forced['value'], plain['value'])
