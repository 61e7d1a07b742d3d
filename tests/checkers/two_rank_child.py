"""Child of tests/test_gpu_two_ranks.py: one RANK of a two-rank job on ONE MI355X.  Strong mode as bench.py --total-frames does
it (distributed.shard_range), the rank's shard through the real engine (match + MLP 3D), the fixed-capacity result blocks
through distributed.all_gather_results.  RCCL refuses two ranks on one device, so the exchange runs on gloo with the blocks
staged through host memory -- the call sequence, the shard arithmetic and the byte packing are the production ones.
argv: rank world total_frames persons out_npz"""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import env, pkg  # noqa: E402


def run_shard(eng, frames, lo, hi, per):
    dmod = pkg('distributed')
    db = eng.to_device(eng.pack(frames[lo:hi]))
    _, persons, n_persons = eng.match(db, want_scores=False)
    poses, _ = eng.mlp3d(db, persons, n_persons)
    eng.sync_status()
    return dmod.pad_to(poses, per), dmod.pad_to(n_persons, per)


def make_frames(e, total, persons):
    syn = pkg('synthetic')
    onp = importlib.import_module('oracle_np') if False else None     # (the oracle is not needed here)
    out = []
    for i in range(total):
        f = syn.make_frame(e.calib, 700 + i, syn.FrameSpec(persons=persons if i % 3 else persons - 2, noise_px=1.0))[0]
        out.append({c: [f[c][0], f[c][1]] for c in f})
    return out


def main():
    rank, world, total, persons, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    e = env('panoptic')
    dmod = pkg('distributed')
    frames = make_frames(e, total, persons)
    lo, hi, per = dmod.shard_range(total, rank, world)
    eng = pkg('pipeline').Engine(e.params, e.calib, max_frames=per, max_persons_per_camera=persons)
    eng.load_gat(*e.gat)
    eng.load_mlp(e.mlp)
    poses, n_persons = run_shard(eng, frames, lo, hi, per)
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
        gp, gn = dmod.all_gather_results(poses.cpu(), n_persons.cpu(), total)
        dist.barrier()
        dist.destroy_process_group()
    else:
        gp, gn = poses[:total].cpu(), n_persons[:total].cpu()
    if rank == 0:
        np.savez(out, poses=gp.numpy(), n_persons=gn.numpy())
    eng.close()


if __name__ == '__main__':
    main()
