"""GPU: the most that one MI355X can say about N > 1: two rank PROCESSES on the one card, strong mode with a
5 x 10 shard each, real engine outputs through distributed.all_gather_results, the gathered block equal bit for bit to a
single-process run over all frames.  (An 8-GPU RCCL run is the driver's; no scaling number is claimed from this.)"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, 'tests', 'checkers', 'two_rank_child.py')


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_two_ranks_on_one_card_gather_the_single_process_bits(tmp_path):
    total, persons = 23, 10                       # 23 frames: shards of 12 and 11 (the second one padded)
    base = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    single = str(tmp_path / 'single.npz')
    r = subprocess.run([sys.executable, CHILD, '0', '1', str(total), str(persons), single], env=base, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    gathered = str(tmp_path / 'gathered.npz')
    procs = [subprocess.Popen([sys.executable, CHILD, str(rank), '2', str(total), str(persons), gathered],
                              env=dict(base, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK='0'), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for rank in (0, 1)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    a, b = np.load(single), np.load(gathered)
    assert a['n_persons'].shape == (total,) and a['n_persons'].sum() > 5 * total
    assert np.array_equal(a['n_persons'], b['n_persons'])
    assert a['poses'].shape == b['poses'].shape and a['poses'].tobytes() == b['poses'].tobytes()
